package quivergpu

/*
#include <stdlib.h>
#include "qv.h"
*/
import "C"

import (
	"errors"
	"fmt"
	"sort"
	"sync"
	"unsafe"

	"github.com/TFMV/quiver/pkg/types"
	"github.com/TFMV/quiver/pkg/vectortypes"
)

// rows is what Index needs from the device side: qv_index_* (one GPU, index.go) or qv_sharded_* (a node, sharded.go).
// Row ids are opaque uint32 (dense on one GPU, one id range per shard on several).
type rows interface {
	add(flat []float32, n int) ([]uint32, error)
	update(row uint32, v []float32) error
	remove(rows []uint32) error
	search(qs []float32, nq, k int) (rows []uint32, dist []float32, count []uint32, err error)
	searchSelected(qs []float32, nq, k int, selected []uint32) ([]uint32, []float32, []uint32, error)
	searchNegative(q, neg []float32, kFetch int) (rows []uint32, dist, negDist []float32, count int, err error)
	distanceRows(q []float32, rows []uint32) ([]float32, error)
	getRows(rows []uint32) ([]float32, error)
	close()
}

// Index implements core.Index and core.BatchIndex (pkg/core/collection.go:78-96) on libqv.  String ids and metadata
// never leave Go; the device sees row numbers, the analogue of hnsw.Node.VectorIndex (pkg/hnsw/hnsw.go:94).
//
// Locking is the reference's: an embedded RWMutex per index (exact.go:25) — Search under RLock, so many goroutines search
// at once (Collection.Search holds only c.RLock, collection.go:647; libqv serves every call from a context of its own),
// mutations under Lock.
type Index struct {
	mu    sync.RWMutex
	dev   rows
	dim   int
	rowOf map[string]uint32 // string id -> device row
	idOf  map[uint32]string // device row -> string id
	free  []uint32          // tombstoned rows, reused by the next Insert (the reference's map frees the entry, exact.go:65)
}

// New replaces hybrid.NewExactIndex (pkg/hybrid/exact.go:29-35) for one GPU.  dim is Collection.Dimension
// (collection.go:105): the dimension lock-in of the first Insert (exact.go:43-47) happens here.
// flags: C.QV_FLAG_BF16_ROWS for collections that serve BatchSearch traffic (+50 % device memory, faster batches).
func New(dim int, m Metric, device int, flags uint64) (*Index, error) {
	defer pinned()() // qv_last_error is thread-local: the failing call and the read of its message stay on one OS thread
	var h *C.qv_index
	if C.qv_index_create(&h, C.uint32_t(dim), C.qv_metric(m), C.int(device), C.uint64_t(flags)) != C.QV_OK {
		return nil, lastErr()
	}
	return newIndex(dim, &oneGPU{h: h, dim: dim}), nil
}

func newIndex(dim int, dev rows) *Index {
	return &Index{dev: dev, dim: dim, rowOf: map[string]uint32{}, idOf: map[uint32]string{}}
}

// Close frees the device memory.  Needs exclusion against every other call (qv_index_destroy).
func (x *Index) Close() { x.mu.Lock(); defer x.mu.Unlock(); x.dev.close() }

// Insert implements core.Index (collection.go:80).  Error strings follow exact.go:45-50.
func (x *Index) Insert(id string, v vectortypes.F32) error {
	x.mu.Lock()
	defer x.mu.Unlock()
	if len(v) != x.dim {
		return fmt.Errorf("vector dimension mismatch: expected %d, got %d", x.dim, len(v))
	}
	if _, ok := x.rowOf[id]; ok {
		return fmt.Errorf("vector with ID %s already exists", id)
	}
	var row uint32
	if n := len(x.free); n > 0 { // overwrite a tombstoned row in place: storage and scan time stay bounded under churn
		row = x.free[n-1]
		if err := x.dev.update(row, v); err != nil {
			return err
		}
		x.free = x.free[:n-1]
	} else {
		got, err := x.dev.add(v, 1) // copies (exact.go:53-56): no Go pointer is retained
		if err != nil {
			return err
		}
		row = got[0]
	}
	x.rowOf[id], x.idOf[row] = row, id
	return nil
}

// InsertBatch implements core.BatchIndex (collection.go:93): one packed upload, all-or-nothing
// (hybrid_index.go:139-216 validates everything first and rolls back on failure; qv_index_add is all-or-nothing).
func (x *Index) InsertBatch(vs map[string]vectortypes.F32) error {
	x.mu.Lock()
	defer x.mu.Unlock()
	ids := make([]string, 0, len(vs))
	for id := range vs {
		ids = append(ids, id)
	}
	sort.Strings(ids) // a Go map has no order; a sorted one makes row numbers (the tie-break) repeatable
	flat := make([]float32, 0, len(vs)*x.dim)
	for _, id := range ids {
		v := vs[id]
		if len(v) != x.dim {
			return fmt.Errorf("vector dimension mismatch: expected %d, got %d", x.dim, len(v))
		}
		if _, ok := x.rowOf[id]; ok {
			return fmt.Errorf("vector with ID %s already exists", id)
		}
		flat = append(flat, v...)
	}
	if len(ids) == 0 {
		return nil
	}
	got, err := x.dev.add(flat, len(ids))
	if err != nil {
		return err
	}
	for i, id := range ids {
		x.rowOf[id], x.idOf[got[i]] = got[i], id
	}
	return nil
}

// Delete implements core.Index; an unknown id is not an error (exact.go:61-70).
func (x *Index) Delete(id string) error {
	x.mu.Lock()
	defer x.mu.Unlock()
	return x.deleteLocked([]string{id})
}

// DeleteBatch implements core.BatchIndex (collection.go:95): one tombstone call.
func (x *Index) DeleteBatch(ids []string) error {
	x.mu.Lock()
	defer x.mu.Unlock()
	return x.deleteLocked(ids)
}

func (x *Index) deleteLocked(ids []string) error {
	dead := make([]uint32, 0, len(ids))
	for _, id := range ids {
		if row, ok := x.rowOf[id]; ok {
			dead = append(dead, row)
		}
	}
	if len(dead) == 0 {
		return nil
	}
	if err := x.dev.remove(dead); err != nil {
		return err
	}
	for _, row := range dead {
		delete(x.rowOf, x.idOf[row])
		delete(x.idOf, row)
	}
	x.free = append(x.free, dead...)
	return nil
}

// Size implements core.Index (exact.go:136-141).
func (x *Index) Size() int { x.mu.RLock(); defer x.mu.RUnlock(); return len(x.rowOf) }

func (x *Index) check(q []float32, k int) (empty bool, err error) { // exact.go:96-106, in that order
	if len(x.rowOf) == 0 {
		return true, nil
	}
	if len(q) != x.dim {
		return false, fmt.Errorf("query dimension mismatch: expected %d, got %d", x.dim, len(q))
	}
	if k <= 0 {
		return false, errors.New("k must be positive")
	}
	return false, nil
}

func (x *Index) results(rows []uint32, dist []float32, n int) []types.BasicSearchResult {
	out := make([]types.BasicSearchResult, n)
	for i := range out {
		out[i] = types.BasicSearchResult{ID: x.idOf[rows[i]], Distance: dist[i]}
	}
	return out
}

// Search implements core.Index (collection.go:84): ascending by distance, ties by device row (a deterministic refinement
// of the reference's unspecified tie order, exact.go:115,124).  k may be Size() — a filtered Collection.Search asks for
// the full ranking (collection.go:679-682), which libqv produces with a device radix sort.
func (x *Index) Search(q vectortypes.F32, k int) ([]types.BasicSearchResult, error) {
	x.mu.RLock()
	defer x.mu.RUnlock()
	empty, err := x.check(q, k)
	if empty || err != nil {
		return []types.BasicSearchResult{}, err
	}
	if k > len(x.rowOf) {
		k = len(x.rowOf) // exact.go:109-111
	}
	rows, dist, count, err := x.dev.search(q, 1, k)
	if err != nil {
		return nil, err
	}
	return x.results(rows, dist, int(count[0])), nil
}

// SearchBatch is HybridIndex.BatchSearch's exact branch (hybrid_index.go:677-811: one goroutine per query) as ONE device
// call: nq queries packed row-major.  Large batches go through the matrix-core filter + exact re-score inside libqv;
// the results are those of nq Search calls.
func (x *Index) SearchBatch(qs []float32, k int) ([][]types.BasicSearchResult, error) {
	x.mu.RLock()
	defer x.mu.RUnlock()
	if x.dim == 0 || len(qs)%x.dim != 0 || len(qs) == 0 {
		return nil, errors.New("no queries provided")
	}
	nq := len(qs) / x.dim
	empty, err := x.check(qs[:x.dim], k)
	if err != nil {
		return nil, err
	}
	out := make([][]types.BasicSearchResult, nq)
	if empty {
		for i := range out {
			out[i] = []types.BasicSearchResult{}
		}
		return out, nil
	}
	if k > len(x.rowOf) {
		k = len(x.rowOf)
	}
	rows, dist, count, err := x.dev.search(qs, nq, k)
	if err != nil {
		return nil, err
	}
	for i := range out {
		out[i] = x.results(rows[i*k:], dist[i*k:], int(count[i]))
	}
	return out, nil
}

// SearchSelected is the filtered Collection.Search (collection.go:679-759) without the full ranking: the caller
// evaluates its metadata filters up front (facets.go:432-460) and passes the matching ids; the k nearest AMONG them come
// back — the same k results the reference reaches by ranking all N rows and keeping the first k matches.
func (x *Index) SearchSelected(q vectortypes.F32, k int, matching []string) ([]types.BasicSearchResult, error) {
	x.mu.RLock()
	defer x.mu.RUnlock()
	empty, err := x.check(q, k)
	if empty || err != nil {
		return []types.BasicSearchResult{}, err
	}
	sel := make([]uint32, 0, len(matching))
	for _, id := range matching {
		if row, ok := x.rowOf[id]; ok {
			sel = append(sel, row)
		}
	}
	rows, dist, count, err := x.dev.searchSelected(q, 1, k, sel)
	if err != nil {
		return nil, err
	}
	return x.results(rows, dist, int(count[0])), nil
}

// SearchWithNegative is the exact branch of HybridIndex.searchWithStrategy with a negative example
// (hybrid_index.go:517-570): fetch retrieveK = max(2k, 30) nearest rows and, for exactly those, distFunc(vector, negative)
// — one device call (qv_index_search_negative) — then score = d - w*d_neg in float32 (:549), stable sort by (score, ID)
// (:552-557), first k (:564-566).
func (x *Index) SearchWithNegative(q, negative vectortypes.F32, weight float32, k int) ([]types.BasicSearchResult, error) {
	x.mu.RLock()
	defer x.mu.RUnlock()
	empty, err := x.check(q, k)
	if empty || err != nil {
		return []types.BasicSearchResult{}, err
	}
	if len(negative) != x.dim {
		return nil, fmt.Errorf("negative example dimension mismatch: expected %d, got %d", x.dim, len(negative))
	}
	retrieveK := 2 * k
	if retrieveK < 30 {
		retrieveK = 30
	}
	if retrieveK > len(x.rowOf) {
		retrieveK = len(x.rowOf)
	}
	rows, dist, neg, n, err := x.dev.searchNegative(q, negative, retrieveK)
	if err != nil {
		return nil, err
	}
	out := make([]types.BasicSearchResult, n)
	for i := 0; i < n; i++ {
		prod := weight * neg[i] // float32, as hybrid_index.go:549
		out[i] = types.BasicSearchResult{ID: x.idOf[rows[i]], Distance: dist[i] - prod}
	}
	sort.SliceStable(out, func(i, j int) bool {
		if out[i].Distance == out[j].Distance {
			return out[i].ID < out[j].ID
		}
		return out[i].Distance < out[j].Distance
	})
	if len(out) > k {
		out = out[:k]
	}
	return out, nil
}

// DistancesTo is the re-rank loop's distFunc(vector, other) for listed ids (hybrid_index.go:536-546; adapter.go:387-415)
// and the neighbour loop of searchLayer (hnsw.go:536-563): one device call for the whole list.
func (x *Index) DistancesTo(other vectortypes.F32, ids []string) ([]float32, error) {
	x.mu.RLock()
	defer x.mu.RUnlock()
	if len(other) != x.dim {
		return nil, fmt.Errorf("query dimension mismatch: expected %d, got %d", x.dim, len(other))
	}
	rows := make([]uint32, len(ids))
	for i, id := range ids {
		row, ok := x.rowOf[id]
		if !ok {
			return nil, fmt.Errorf("vector with ID %s not found", id)
		}
		rows[i] = row
	}
	return x.dev.distanceRows(other, rows)
}

// Vector returns a copy of the stored vector (hybrid_index.go:537 reads idx.vectors[id]).
func (x *Index) Vector(id string) (vectortypes.F32, bool, error) {
	x.mu.RLock()
	defer x.mu.RUnlock()
	row, ok := x.rowOf[id]
	if !ok {
		return nil, false, nil
	}
	v, err := x.dev.getRows([]uint32{row})
	return v, err == nil, err
}

// ---------------------------------------------------------------------------------------------- one GPU: qv_index_*

type oneGPU struct {
	h   *C.qv_index
	dim int
}

func f32p(s []float32) *C.float {
	if len(s) == 0 {
		return nil
	}
	return (*C.float)(unsafe.Pointer(&s[0]))
}
func u32p(s []uint32) *C.uint32_t {
	if len(s) == 0 {
		return nil
	}
	return (*C.uint32_t)(unsafe.Pointer(&s[0]))
}

func (d *oneGPU) close() { C.qv_index_destroy(d.h); d.h = nil }

func (d *oneGPU) add(flat []float32, n int) ([]uint32, error) {
	defer pinned()() // qv_last_error is thread-local: the failing call and the read of its message stay on one OS thread
	var first C.uint32_t
	if C.qv_index_add(d.h, f32p(flat), C.uint32_t(n), &first) != C.QV_OK {
		return nil, lastErr()
	}
	out := make([]uint32, n)
	for i := range out {
		out[i] = uint32(first) + uint32(i)
	}
	return out, nil
}

func (d *oneGPU) update(row uint32, v []float32) error {
	defer pinned()() // qv_last_error is thread-local: the failing call and the read of its message stay on one OS thread
	if C.qv_index_update(d.h, C.uint32_t(row), f32p(v)) != C.QV_OK {
		return lastErr()
	}
	return nil
}

func (d *oneGPU) remove(rows []uint32) error {
	defer pinned()() // qv_last_error is thread-local: the failing call and the read of its message stay on one OS thread
	if C.qv_index_remove(d.h, u32p(rows), C.uint32_t(len(rows))) != C.QV_OK {
		return lastErr()
	}
	return nil
}

func (d *oneGPU) search(qs []float32, nq, k int) ([]uint32, []float32, []uint32, error) {
	defer pinned()() // qv_last_error is thread-local: the failing call and the read of its message stay on one OS thread
	rows, dist, count := make([]uint32, nq*k), make([]float32, nq*k), make([]uint32, nq)
	// qv_index_search routes big batches to the matrix-core filter + exact re-score by itself (identical results)
	if C.qv_index_search(d.h, f32p(qs), C.uint32_t(nq), C.uint32_t(k), u32p(rows), f32p(dist), u32p(count)) != C.QV_OK {
		return nil, nil, nil, lastErr()
	}
	return rows, dist, count, nil
}

func (d *oneGPU) searchSelected(qs []float32, nq, k int, selected []uint32) ([]uint32, []float32, []uint32, error) {
	defer pinned()() // qv_last_error is thread-local: the failing call and the read of its message stay on one OS thread
	words := (int(C.qv_index_rows(d.h)) + 63) / 64
	mask := make([]uint64, words+1)
	for _, r := range selected {
		mask[r>>6] |= 1 << (r & 63)
	}
	rows, dist, count := make([]uint32, nq*k), make([]float32, nq*k), make([]uint32, nq)
	if C.qv_index_search_masked(d.h, f32p(qs), C.uint32_t(nq), C.uint32_t(k), (*C.uint64_t)(unsafe.Pointer(&mask[0])),
		u32p(rows), f32p(dist), u32p(count)) != C.QV_OK {
		return nil, nil, nil, lastErr()
	}
	return rows, dist, count, nil
}

func (d *oneGPU) searchNegative(q, neg []float32, kFetch int) ([]uint32, []float32, []float32, int, error) {
	defer pinned()() // qv_last_error is thread-local: the failing call and the read of its message stay on one OS thread
	rows, dist, nd := make([]uint32, kFetch), make([]float32, kFetch), make([]float32, kFetch)
	var n C.uint32_t
	if C.qv_index_search_negative(d.h, f32p(q), f32p(neg), C.uint32_t(kFetch), u32p(rows), f32p(dist), f32p(nd), &n) != C.QV_OK {
		return nil, nil, nil, 0, lastErr()
	}
	return rows, dist, nd, int(n), nil
}

func (d *oneGPU) distanceRows(q []float32, rows []uint32) ([]float32, error) {
	defer pinned()() // qv_last_error is thread-local: the failing call and the read of its message stay on one OS thread
	out := make([]float32, len(rows))
	if C.qv_distance_rows(d.h, f32p(q), u32p(rows), C.uint32_t(len(rows)), f32p(out)) != C.QV_OK {
		return nil, lastErr()
	}
	return out, nil
}

func (d *oneGPU) getRows(rows []uint32) ([]float32, error) {
	defer pinned()() // qv_last_error is thread-local: the failing call and the read of its message stay on one OS thread
	out := make([]float32, len(rows)*d.dim)
	if C.qv_index_get_rows(d.h, u32p(rows), C.uint32_t(len(rows)), f32p(out)) != C.QV_OK {
		return nil, lastErr()
	}
	return out, nil
}
