package quivergpu

/*
#include "qv.h"
*/
import "C"

import (
	"errors"
	"sync"
	"unsafe"
)

// Graph is the device-resident HNSW of libqv (qv_graph_*): hnsw.HNSW's Nodes / Connections (pkg/hnsw/hnsw.go:44-85) in
// HBM over the rows of ONE-GPU row storage created with C.QV_FLAG_ROWMAJOR (node index == device row, the analogue of
// `newNodeIdx := uint32(len(h.Nodes))`, hnsw.go:279).  Whole queries are walked on the GPU, one wavefront per query, with
// the reference's searchLayer semantics (same heaps, same admission order, bit-identical distances, hnsw.go:471-580).
//
// A per-hop qv_distance_rows call from Go would be launch-bound (~10 us per hop against 3 KB of useful reads), so the
// offload is by whole queries: SearchBatch.  The level law and its RNG (randomLevel, hnsw.go:716-738; seeded from the wall
// clock, hnsw.go:248) stay in Go: the caller draws one level per inserted node, in node order.
type Graph struct {
	mu       sync.RWMutex // hnsw.go:58: searches shared, Insert exclusive
	g        *C.qv_graph
	idx      *C.qv_index
	dim      int
	EfSearch int // hnsw.go:232-234 default 100
}

// GraphConfig is hnsw.Config (hnsw.go:27-41); zero values take the reference's defaults 16 / 2M / 200 / 100 (hnsw.go:223-237).
type GraphConfig struct {
	M, MaxM0, EfConstruction, EfSearch int
}

// NewGraph creates the vector storage (a qv_index with a row-major copy for the per-hop gathers) and an empty graph over it.
func NewGraph(dim int, m Metric, device int, capacity int, cfg GraphConfig) (*Graph, error) {
	defer pinned()() // qv_last_error is thread-local: the failing call and the read of its message stay on one OS thread
	var idx *C.qv_index
	if C.qv_index_create(&idx, C.uint32_t(dim), C.qv_metric(m), C.int(device), C.QV_FLAG_ROWMAJOR) != C.QV_OK {
		return nil, lastErr()
	}
	var g *C.qv_graph
	if C.qv_graph_create_empty(&g, idx, C.uint32_t(capacity), C.uint32_t(cfg.M), C.uint32_t(cfg.MaxM0), C.uint32_t(cfg.EfConstruction)) != C.QV_OK {
		err := lastErr()
		C.qv_index_destroy(idx)
		return nil, err
	}
	ef := cfg.EfSearch
	if ef <= 0 {
		ef = 100
	}
	return &Graph{g: g, idx: idx, dim: dim, EfSearch: ef}, nil
}

func (h *Graph) Close() {
	h.mu.Lock()
	defer h.mu.Unlock()
	C.qv_graph_destroy(h.g)
	C.qv_index_destroy(h.idx)
	h.g, h.idx = nil, nil
}

// InsertBatch is a loop of hnsw.HNSW.Insert (hnsw.go:266-334; connectNode :337-468) for n vectors packed row-major,
// connected on the device in batches: every node of a batch searches the graph as it was before the batch, links are then
// applied as if node by node in index order — the deterministic form of the reference's own concurrent Inserts (it drops its
// lock before connectNode, hnsw.go:313-315).  levels[i] = randomLevel() of vector i, drawn by the caller in order.
// batchMax = 1 reproduces n sequential Inserts exactly; 0 = the library's default (16384, ramped up from 1).
// Returns the node index of the first inserted vector.
func (h *Graph) InsertBatch(flat []float32, levels []int8, batchMax int) (uint32, error) {
	defer pinned()() // qv_last_error is thread-local: the failing call and the read of its message stay on one OS thread
	h.mu.Lock()
	defer h.mu.Unlock()
	n := len(levels)
	if n == 0 {
		return 0, nil
	}
	if len(flat) != n*h.dim {
		return 0, errors.New("vector dimensions do not match") // adapter.go:171 ErrDimensionMismatch
	}
	var first C.uint32_t
	if C.qv_index_add(h.idx, f32p(flat), C.uint32_t(n), &first) != C.QV_OK {
		return 0, lastErr()
	}
	if C.qv_graph_insert(h.g, first, C.uint32_t(n), (*C.int8_t)(unsafe.Pointer(&levels[0])), C.uint32_t(batchMax), 16) != C.QV_OK {
		err := lastErr()
		rows := make([]uint32, n) // hnsw.go:316-322: a failed connect keeps the slots as tombstones
		for i := range rows {
			rows[i] = uint32(first) + uint32(i)
		}
		C.qv_index_remove(h.idx, u32p(rows), C.uint32_t(n))
		return 0, err
	}
	return uint32(first), nil
}

// Search is hnsw.HNSW.Search (hnsw.go:602-713) for ONE query — what the reference's callers send, one goroutine per request under a
// read lock (hnsw.go:602-606; adapter.go:253-279).  It is a batch of one: libqv lets concurrent calls on one graph share traversal
// batches (include/qv.h, qv_graph_search), so a thousand goroutines calling this cost a few batches, not a thousand walks one after
// another; a lone caller is served at once.
func (h *Graph) Search(q []float32, k int) ([]GraphResult, error) {
	res, err := h.SearchBatch(q, k)
	if err != nil || len(res) == 0 {
		return nil, err
	}
	return res[0], nil
}

// GraphResult is hnsw.Result (hnsw.go:87-95) without the string id: the caller maps VectorIndex to its ids.
type GraphResult struct {
	VectorIndex uint32
	Distance    float32
}

// SearchBatch is hnsw.HNSW.Search (hnsw.go:602-713) for nq queries packed row-major: greedy descent through the upper
// levels, searchLayer(max(EfSearch, k)) on level 0, first k.  A query whose graph search returns fewer than k results
// (hnsw.go:676) is completed like the reference's brute-force top-up (hnsw.go:676-710) by ONE exact scan call for all
// such queries — the exact top-k over all live nodes under (distance, node) order.
func (h *Graph) SearchBatch(qs []float32, k int) ([][]GraphResult, error) {
	defer pinned()() // qv_last_error is thread-local: the failing call and the read of its message stay on one OS thread
	h.mu.RLock()
	defer h.mu.RUnlock()
	if k <= 0 {
		return nil, errors.New("k must be positive") // hnsw.go:610-612
	}
	if h.dim == 0 || len(qs) == 0 || len(qs)%h.dim != 0 {
		return nil, errors.New("vector dimensions do not match")
	}
	nq := len(qs) / h.dim
	out := make([][]GraphResult, nq)
	live := int(C.qv_index_size(h.idx))
	if live == 0 {
		return out, nil // hnsw.go:606-608
	}
	if k > live {
		k = live // hnsw.go:615-617 (clamped to the live nodes: the top-up below cannot return more)
	}
	rows, dist, count := make([]uint32, nq*k), make([]float32, nq*k), make([]uint32, nq)
	if C.qv_graph_search(h.g, f32p(qs), C.uint32_t(nq), C.uint32_t(k), C.uint32_t(h.EfSearch), u32p(rows), f32p(dist), u32p(count), nil) != C.QV_OK {
		return nil, lastErr()
	}
	var under []int
	for q := 0; q < nq; q++ {
		if int(count[q]) == k {
			r := make([]GraphResult, k)
			for i := range r {
				r[i] = GraphResult{rows[q*k+i], dist[q*k+i]}
			}
			out[q] = r
		} else {
			under = append(under, q)
		}
	}
	if len(under) > 0 { // hnsw.go:676-710
		packed := make([]float32, 0, len(under)*h.dim)
		for _, q := range under {
			packed = append(packed, qs[q*h.dim:(q+1)*h.dim]...)
		}
		r2, d2, c2 := make([]uint32, len(under)*k), make([]float32, len(under)*k), make([]uint32, len(under))
		if C.qv_index_search(h.idx, f32p(packed), C.uint32_t(len(under)), C.uint32_t(k), u32p(r2), f32p(d2), u32p(c2)) != C.QV_OK {
			return nil, lastErr()
		}
		for j, q := range under {
			r := make([]GraphResult, int(c2[j]))
			for i := range r {
				r[i] = GraphResult{r2[j*k+i], d2[j*k+i]}
			}
			out[q] = r
		}
	}
	return out, nil
}

// Delete tombstones a node (hnsw.go:829 Nodes[idx] = nil).  The device graph keeps walking through it until the caller
// re-uploads the adjacency its Go-side Delete produced (hnsw.go:741-842 unlinks the node); see FromAdjacency.
func (h *Graph) Delete(node uint32) error {
	defer pinned()() // qv_last_error is thread-local: the failing call and the read of its message stay on one OS thread
	h.mu.Lock()
	defer h.mu.Unlock()
	if C.qv_index_remove(h.idx, &[]C.uint32_t{C.uint32_t(node)}[0], 1) != C.QV_OK {
		return lastErr()
	}
	return nil
}

// Adjacency is the flat form of hnsw.Node.Level / Connections (hnsw.go:44-56) that qv_graph_create takes and qv_graph_export
// returns: Levels[i] = -1 for a nil node; level 0 as degree + MaxM0 links per node; each (node, level >= 1) as a block of
// 1 + M words (degree, links), node i's first block at UpOff[i].
type Adjacency struct {
	Levels        []int8
	L0Deg, L0     []uint32
	UpOff, Up     []uint32
	MaxM0, M      int
	Entry         uint32
	CurrentLevel  int
}

// Export copies the device graph back (what HNSW.Nodes[i].Connections holds) for Delete and persistence on the Go side.
func (h *Graph) Export() (*Adjacency, error) {
	defer pinned()() // qv_last_error is thread-local: the failing call and the read of its message stay on one OS thread
	h.mu.RLock()
	defer h.mu.RUnlock()
	var n, nb, m0, m, ep C.uint32_t
	var lvl C.int
	if C.qv_graph_info(h.g, &n, &nb, &m0, &m, &ep, &lvl) != C.QV_OK {
		return nil, lastErr()
	}
	a := &Adjacency{Levels: make([]int8, n), L0Deg: make([]uint32, n), L0: make([]uint32, int(n)*int(m0)), UpOff: make([]uint32, n),
		Up: make([]uint32, (int(nb)+1)*(1+int(m))), MaxM0: int(m0), M: int(m), Entry: uint32(ep), CurrentLevel: int(lvl)}
	if n == 0 {
		return a, nil
	}
	if C.qv_graph_export(h.g, (*C.int8_t)(unsafe.Pointer(&a.Levels[0])), u32p(a.L0Deg), u32p(a.L0), u32p(a.UpOff), u32p(a.Up)) != C.QV_OK {
		return nil, lastErr()
	}
	a.Up = a.Up[:int(nb)*(1+int(m))]
	return a, nil
}

// FromAdjacency uploads a graph built (or edited: Delete) on the Go side over the vectors already in this Graph's storage,
// replacing the device graph; MakeBuildable lets InsertBatch extend it afterwards (it scores every link once).
func (h *Graph) FromAdjacency(a *Adjacency, efConstruction int) error {
	defer pinned()() // qv_last_error is thread-local: the failing call and the read of its message stay on one OS thread
	h.mu.Lock()
	defer h.mu.Unlock()
	n := len(a.Levels)
	if n == 0 {
		return errors.New("graph has no live node")
	}
	up := a.Up
	if len(up) == 0 {
		up = make([]uint32, 1+a.M)
	}
	var g *C.qv_graph
	if C.qv_graph_create(&g, h.idx, C.uint32_t(n), (*C.int8_t)(unsafe.Pointer(&a.Levels[0])), C.uint32_t(a.MaxM0), C.uint32_t(a.M),
		u32p(a.L0Deg), u32p(a.L0), u32p(a.UpOff), u32p(up), C.uint32_t(len(up)/(1+a.M)), C.uint32_t(a.Entry), C.int(a.CurrentLevel)) != C.QV_OK {
		return lastErr()
	}
	if C.qv_graph_make_buildable(g, C.uint32_t(efConstruction)) != C.QV_OK {
		err := lastErr()
		C.qv_graph_destroy(g)
		return err
	}
	C.qv_graph_destroy(h.g)
	h.g = g
	return nil
}

// Replicas is "HNSW across the GPUs of a node" (SURVEY.md 8e: graph traversal is sequentially dependent, so the graph does
// not shard — replicas only): one complete Graph per device, every InsertBatch applied to all of them (the build is
// deterministic, so the replicas are identical), the queries of a SearchBatch cut into one contiguous slice per device and
// walked concurrently.
type Replicas struct {
	G []*Graph
}

func NewReplicas(dim int, m Metric, devices []int, capacity int, cfg GraphConfig) (*Replicas, error) {
	r := &Replicas{}
	for _, d := range devices {
		g, err := NewGraph(dim, m, d, capacity, cfg)
		if err != nil {
			r.Close()
			return nil, err
		}
		r.G = append(r.G, g)
	}
	return r, nil
}

func (r *Replicas) Close() {
	for _, g := range r.G {
		g.Close()
	}
}

func (r *Replicas) InsertBatch(flat []float32, levels []int8, batchMax int) (uint32, error) {
	var first uint32
	errs := make([]error, len(r.G))
	firsts := make([]uint32, len(r.G))
	var wg sync.WaitGroup
	for i, g := range r.G {
		wg.Add(1)
		go func(i int, g *Graph) { defer wg.Done(); firsts[i], errs[i] = g.InsertBatch(flat, levels, batchMax) }(i, g)
	}
	wg.Wait()
	for i, err := range errs {
		if err != nil {
			return 0, err
		}
		if i > 0 && firsts[i] != firsts[0] {
			return 0, errors.New("replicas diverged")
		}
		first = firsts[i]
	}
	return first, nil
}

func (r *Replicas) SearchBatch(qs []float32, k int) ([][]GraphResult, error) {
	if len(r.G) == 0 {
		return nil, errors.New("no replicas")
	}
	dim := r.G[0].dim
	nq := len(qs) / dim
	out := make([][]GraphResult, nq)
	errs := make([]error, len(r.G))
	var wg sync.WaitGroup
	for i, g := range r.G {
		lo, hi := i*nq/len(r.G), (i+1)*nq/len(r.G)
		if lo == hi {
			continue
		}
		wg.Add(1)
		go func(i, lo, hi int, g *Graph) {
			defer wg.Done()
			res, err := g.SearchBatch(qs[lo*dim:hi*dim], k)
			errs[i] = err
			copy(out[lo:hi], res)
		}(i, lo, hi, g)
	}
	wg.Wait()
	for _, err := range errs {
		if err != nil {
			return nil, err
		}
	}
	return out, nil
}
