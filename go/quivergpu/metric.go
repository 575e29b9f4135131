package quivergpu

/*
#cgo CFLAGS: -I${SRCDIR}/../../include
#cgo LDFLAGS: -L${SRCDIR}/../../quiver_amd/lib -lqv -Wl,-rpath,${SRCDIR}/../../quiver_amd/lib
#include "qv.h"
*/
import "C"

import (
	"errors"
	"fmt"
	"runtime"
	"unsafe"

	"github.com/TFMV/quiver/pkg/hnsw"
	"github.com/TFMV/quiver/pkg/vectortypes"
)

// Metric is a qv_metric (include/qv.h): 0..4 restate pkg/vectortypes/distances.go:12-104, 5..7 pkg/hnsw/adapter.go:105-167.
type Metric int

const (
	Cosine     Metric = C.QV_COSINE
	L2         Metric = C.QV_L2
	L2Squared  Metric = C.QV_L2SQ
	Dot        Metric = C.QV_DOT
	L1         Metric = C.QV_L1
	CosineF32  Metric = C.QV_COSINE_F32
	L2F32      Metric = C.QV_L2_F32
	DotF32     Metric = C.QV_DOT_F32
	L2SqF64    Metric = C.QV_L2SQ_F64
)

// lastErr reads libqv's message for the call that just failed.  The message is THREAD-local and a goroutine may be moved to
// another OS thread between two cgo calls, so every function that can reach lastErr pins itself first (pinned): the failing call
// and this read then run on one thread.
func lastErr() error { return errors.New(C.GoString(C.qv_last_error())) }

// pinned locks the calling goroutine to its OS thread until the returned function runs: `defer pinned()()`.
func pinned() func() {
	runtime.LockOSThread()
	return runtime.UnlockOSThread
}

// MetricOf identifies a vectortypes.DistanceFunc the way DB.CreateCollection does — by comparing function pointers
// printed with %p (pkg/core/db.go:326-334, 359-367).  ok == false: an arbitrary closure, which cannot be offloaded; the
// caller keeps the Go index.
func MetricOf(f vectortypes.DistanceFunc) (Metric, bool) {
	p := fmt.Sprintf("%p", f)
	switch p {
	case fmt.Sprintf("%p", vectortypes.CosineDistance):
		return Cosine, true
	case fmt.Sprintf("%p", vectortypes.EuclideanDistance):
		return L2, true
	case fmt.Sprintf("%p", vectortypes.SquaredEuclideanDistance):
		return L2Squared, true
	case fmt.Sprintf("%p", vectortypes.DotProductDistance):
		return Dot, true
	case fmt.Sprintf("%p", vectortypes.ManhattanDistance):
		return L1, true
	}
	return Cosine, false
}

// MetricOfHNSW does the same for the pkg/hnsw functions a reloaded collection uses (pkg/core/db.go:181-188).
func MetricOfHNSW(f hnsw.DistanceFunction) (Metric, bool) {
	p := fmt.Sprintf("%p", f)
	switch p {
	case fmt.Sprintf("%p", hnsw.CosineDistanceFunc):
		return CosineF32, true
	case fmt.Sprintf("%p", hnsw.EuclideanDistanceFunc):
		return L2F32, true
	case fmt.Sprintf("%p", hnsw.DotProductDistanceFunc):
		return DotF32, true
	}
	return CosineF32, false
}

// MetricByType mirrors vectortypes.GetDistanceFuncByType (pkg/vectortypes/types.go:36-49): unknown types are cosine.
func MetricByType(t vectortypes.DistanceType) Metric {
	switch t {
	case vectortypes.Euclidean:
		return L2
	case vectortypes.DotProduct:
		return Dot
	case vectortypes.Manhattan:
		return L1
	}
	return Cosine
}

// DistanceFunc returns a vectortypes.DistanceFunc (pkg/vectortypes/surface.go:8) computed by libqv on the HOST
// (qv_distance_pair: the kernels' own per-pair arithmetic compiled for the CPU — a per-pair call costs the reference 78 ns,
// final_bench.txt:47, and no device round trip can serve that).  Bit-identical to what the device scans return for the
// same pair.  Panics on a length mismatch like distances.go:13-15.  Batches of pairs belong to DistancePairs.
func DistanceFunc(m Metric) vectortypes.DistanceFunc {
	return func(a, b vectortypes.F32) float32 {
		if len(a) != len(b) {
			panic("vectors must have the same length")
		}
		var out C.float
		var pa, pb *C.float
		if len(a) > 0 {
			pa, pb = (*C.float)(unsafe.Pointer(&a[0])), (*C.float)(unsafe.Pointer(&b[0]))
		}
		if C.qv_distance_pair(C.qv_metric(m), pa, pb, C.uint32_t(len(a)), &out) != C.QV_OK {
			// (only an unknown metric or a null vector gets here; the per-pair path does not pin its thread for the message)
			panic(fmt.Sprintf("qv_distance_pair failed for metric %d on %d dimensions", int(m), len(a)))
		}
		return float32(out)
	}
}

// HNSWDistanceFunction is the (float32, error) form pkg/hnsw wants (hnsw.go:14): ErrDimensionMismatch instead of a panic
// (adapter.go:106-108).
func HNSWDistanceFunction(m Metric) hnsw.DistanceFunction {
	return func(a, b []float32) (float32, error) {
		if len(a) != len(b) {
			return 0, hnsw.ErrDimensionMismatch
		}
		var out C.float
		var pa, pb *C.float
		if len(a) > 0 {
			pa, pb = (*C.float)(unsafe.Pointer(&a[0])), (*C.float)(unsafe.Pointer(&b[0]))
		}
		if C.qv_distance_pair(C.qv_metric(m), pa, pb, C.uint32_t(len(a)), &out) != C.QV_OK {
			return 0, fmt.Errorf("qv_distance_pair failed for metric %d on %d dimensions", int(m), len(a))
		}
		return float32(out), nil
	}
}

// DistancePairs evaluates n independent pairs a[i], b[i] (each dim long, packed row-major) on the device.
func DistancePairs(m Metric, a, b []float32, dim, device int) ([]float32, error) {
	defer pinned()() // qv_last_error is thread-local: the failing call and the read of its message stay on one OS thread
	if dim <= 0 || len(a) != len(b) || len(a)%dim != 0 {
		return nil, errors.New("vectors must have the same length")
	}
	n := len(a) / dim
	out := make([]float32, n)
	if n == 0 {
		return out, nil
	}
	if C.qv_distance_pairs(C.qv_metric(m), (*C.float)(unsafe.Pointer(&a[0])), (*C.float)(unsafe.Pointer(&b[0])),
		C.uint32_t(n), C.uint32_t(dim), (*C.float)(unsafe.Pointer(&out[0])), C.int(device)) != C.QV_OK {
		return nil, lastErr()
	}
	return out, nil
}
