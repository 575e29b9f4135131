// Package quivergpu puts libqv — the MI355X (gfx950) similarity-search hot path — behind Quiver's own index seam.
//
// It is NEW Go code a Quiver maintainer adds to the module github.com/TFMV/quiver; nothing in the reference crosses a
// language boundary today.  The seam is pkg/core.Index + pkg/core.BatchIndex (pkg/core/collection.go:78-96), the two
// interfaces the reference already substitutes in its own tests (MockIndex / MockBatchIndex, collection_test.go:13-80);
// the construction site is DB.CreateCollection (pkg/core/db.go:312-374), which today builds hybrid.NewHybridIndex or
// hnsw.NewAdapter and would build quivergpu.New / quivergpu.NewSharded instead when the distance function is one of the
// eight libqv restates (metric.go).  Everything above the seam — Collection.Add / AddBatch / Search / FluentSearch,
// filters, facets, persistence, REST — is unchanged.
//
//	index.go    Index: core.Index + core.BatchIndex over ONE GPU (qv_index_*), plus the three calls the seam's callers
//	            can use to avoid work: SearchBatch (hybrid BatchSearch), SearchSelected (filtered Collection.Search
//	            without the full ranking), SearchWithNegative (hybrid_index.go:517-570)
//	sharded.go  Sharded: the same surface over the GPUs of a node (qv_sharded_*: one row shard per device, one RCCL
//	            all-gather per search)
//	graph.go    Graph: the device-resident HNSW (qv_graph_*): build / insert / batched search / export
//	metric.go   which vectortypes / hnsw distance function is which qv_metric; the per-pair DistanceFunc
//
// Build: `make -C quiver_amd/csrc` produces quiver_amd/lib/libqv.so; Quiver's Dockerfile already builds with
// CGO_ENABLED=1 (Dockerfile:20).  The authoring image of this repository has no Go toolchain, so these files are checked
// against include/qv.h by tests/test_go_binding.py (every C.qv_* call names a declared function with the declared number
// of arguments) rather than by the Go compiler.
//
// cgo rules observed throughout: libqv copies on add and retains no pointer after a call returns; every slice passed
// down is pinned only for the duration of the call; outputs are caller-allocated Go slices.
package quivergpu
