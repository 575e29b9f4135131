package quivergpu

/*
#include "qv.h"
*/
import "C"

// NewSharded is New over the GPUs of a node: one row shard per listed device behind ONE qv_sharded handle — every
// shard scans on its own GPU, one RCCL all-gather (xGMI) carries the per-shard result lists, the first device merges
// under the same (distance, row) order (SURVEY.md 8e; the reference is one process on CPU cores and has no counterpart).
// The returned *Index is the same type with the same methods as New's: core.Index + core.BatchIndex, any k, filtered
// search, negative examples.  Row ids are `shard*span + local row` and stay stable as shards grow.
//
// flags: C.QV_FLAG_BF16_ROWS passes through to every shard; C.QV_SHARDED_PEER_COPY replaces the collective with
// point-to-point copies into the first device (and allows a device to be listed more than once).
func NewSharded(dim int, m Metric, devices []int, flags uint64) (*Index, error) {
	defer pinned()() // qv_last_error is thread-local: the failing call and the read of its message stay on one OS thread
	devs := make([]C.int, len(devices))
	for i, d := range devices {
		devs[i] = C.int(d)
	}
	var h *C.qv_sharded
	var p *C.int
	if len(devs) > 0 {
		p = &devs[0]
	}
	if C.qv_sharded_create(&h, C.uint32_t(dim), C.qv_metric(m), p, C.int(len(devs)), C.uint64_t(flags)) != C.QV_OK {
		return nil, lastErr()
	}
	return newIndex(dim, &node{h: h, dim: dim}), nil
}

// RuntimeInfo names the HIP runtime and the RCCL build the process bound (for reports of multi-GPU runs).
func RuntimeInfo() string {
	buf := make([]C.char, 1024)
	if C.qv_runtime_info(&buf[0], C.size_t(len(buf))) != C.QV_OK {
		return ""
	}
	return C.GoString(&buf[0])
}

type node struct {
	h   *C.qv_sharded
	dim int
}

func (d *node) close() { C.qv_sharded_destroy(d.h); d.h = nil }

func (d *node) add(flat []float32, n int) ([]uint32, error) {
	defer pinned()() // qv_last_error is thread-local: the failing call and the read of its message stay on one OS thread
	out := make([]uint32, n)
	if C.qv_sharded_add(d.h, f32p(flat), C.uint32_t(n), u32p(out)) != C.QV_OK { // all-or-nothing; out[i] = global row of vector i
		return nil, lastErr()
	}
	return out, nil
}

func (d *node) update(row uint32, v []float32) error {
	defer pinned()() // qv_last_error is thread-local: the failing call and the read of its message stay on one OS thread
	if C.qv_sharded_update(d.h, C.uint32_t(row), f32p(v)) != C.QV_OK {
		return lastErr()
	}
	return nil
}

func (d *node) remove(rows []uint32) error {
	defer pinned()() // qv_last_error is thread-local: the failing call and the read of its message stay on one OS thread
	if C.qv_sharded_remove(d.h, u32p(rows), C.uint32_t(len(rows))) != C.QV_OK {
		return lastErr()
	}
	return nil
}

func (d *node) search(qs []float32, nq, k int) ([]uint32, []float32, []uint32, error) {
	defer pinned()() // qv_last_error is thread-local: the failing call and the read of its message stay on one OS thread
	rows, dist, count := make([]uint32, nq*k), make([]float32, nq*k), make([]uint32, nq)
	if C.qv_sharded_search(d.h, f32p(qs), C.uint32_t(nq), C.uint32_t(k), u32p(rows), f32p(dist), u32p(count)) != C.QV_OK {
		return nil, nil, nil, lastErr()
	}
	return rows, dist, count, nil
}

func (d *node) searchSelected(qs []float32, nq, k int, selected []uint32) ([]uint32, []float32, []uint32, error) {
	defer pinned()() // qv_last_error is thread-local: the failing call and the read of its message stay on one OS thread
	rows, dist, count := make([]uint32, nq*k), make([]float32, nq*k), make([]uint32, nq)
	if C.qv_sharded_search_masked(d.h, f32p(qs), C.uint32_t(nq), C.uint32_t(k), u32p(selected), C.uint32_t(len(selected)),
		u32p(rows), f32p(dist), u32p(count)) != C.QV_OK {
		return nil, nil, nil, lastErr()
	}
	return rows, dist, count, nil
}

func (d *node) searchNegative(q, neg []float32, kFetch int) ([]uint32, []float32, []float32, int, error) {
	defer pinned()() // qv_last_error is thread-local: the failing call and the read of its message stay on one OS thread
	rows, dist, nd := make([]uint32, kFetch), make([]float32, kFetch), make([]float32, kFetch)
	var n C.uint32_t
	if C.qv_sharded_search_negative(d.h, f32p(q), f32p(neg), C.uint32_t(kFetch), u32p(rows), f32p(dist), f32p(nd), &n) != C.QV_OK {
		return nil, nil, nil, 0, lastErr()
	}
	return rows, dist, nd, int(n), nil
}

func (d *node) distanceRows(q []float32, rows []uint32) ([]float32, error) {
	defer pinned()() // qv_last_error is thread-local: the failing call and the read of its message stay on one OS thread
	out := make([]float32, len(rows))
	if C.qv_sharded_distance_rows(d.h, f32p(q), u32p(rows), C.uint32_t(len(rows)), f32p(out)) != C.QV_OK {
		return nil, lastErr()
	}
	return out, nil
}

func (d *node) getRows(rows []uint32) ([]float32, error) {
	defer pinned()() // qv_last_error is thread-local: the failing call and the read of its message stay on one OS thread
	out := make([]float32, len(rows)*d.dim)
	if C.qv_sharded_get_rows(d.h, u32p(rows), C.uint32_t(len(rows)), f32p(out)) != C.QV_OK {
		return nil, lastErr()
	}
	return out, nil
}
