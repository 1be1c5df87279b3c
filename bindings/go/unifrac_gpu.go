// unifrac_gpu.go -- the cgo shim that puts libfrackyfrac_amd behind frcfrc's pairwise stage.
//
// Drop this file next to frcfrc/unifrac.go (package main) and have unifrac() (unifrac.go:123)
// return the sequence of unifracDistsGPU(nodes, treeDists, weighted) instead of
// unifracDists(...); INTEGRATION.md section 1 shows the two-line patch.  It keeps the reference's
// shape: an iter.Seq[float64] in common.IterPairs order that is LAZY -- nothing is converted,
// staged or computed until it is ranged over, and a sequence that is never ranged costs nothing
// and owns nothing (unifrac.go:209-211) -- and that stops computing when the consumer stops
// (unifrac.go:221-226): the library walks the pair space in sub-shards and the callback's
// return value ends the walk.  Source only: this image has no Go toolchain.  The call sequence
// is exercised, call for call, by tests/harness/go_shim_sequence.c on the reference's golden
// files, and tests/test_go_shim_harness.py checks this file against cgo's pointer rule.
//
// cgo pointer passing ("Go code may pass a Go pointer to C provided the Go memory to which it
// points does not contain any Go pointers"): every pointer below is a TOP-LEVEL argument that
// points at pointer-free Go memory (slices of numbers, a C.ff_options, a cgo.Handle) and is only
// borrowed for the duration of the call.  No C struct holding Go pointers (a C.ff_problem filled
// from slices) is ever built -- that is what the *_csr entry points are for.
package main

/*
#cgo CFLAGS: -I${SRCDIR}/../../include
#cgo LDFLAGS: -L${SRCDIR}/../../frackyfrac_amd/lib -lfrackyfrac_amd
#include <stdint.h>
#include "frackyfrac_amd.h"

// ffDeliver is the exported Go function below.  Declared the way cgo itself declares exported
// functions in _cgo_export.h (cgo drops const: a `const double *` here would be a conflicting
// declaration); it is passed as an ff_dists_fn.
extern int ffDeliver(void *user, int64_t slotBegin, double *dists, int64_t n);
extern int ffWriteText(void *user, char *text, size_t n);
*/
import "C"

import (
	"fmt"
	"io"
	"iter"
	"runtime/cgo"
	"unsafe"
)

// gpuSeq is what one ranging of the sequence shares with its callback.
type gpuSeq struct {
	yield func(float64) bool
}

// ffDeliver receives one finished piece of the pair space: dists[0:n] are the distances of the
// global slots slotBegin .. slotBegin+n-1 (library-owned host memory, valid during the call).
// The library calls it on the thread that called ff_unifrac_dists_stream_csr, i.e. on the
// goroutine that ranges over the sequence, so calling yield from here is legal.
//
//export ffDeliver
func ffDeliver(user unsafe.Pointer, slotBegin C.int64_t, dists *C.double, n C.int64_t) C.int {
	s := (*(*cgo.Handle)(user)).Value().(*gpuSeq)
	for _, d := range unsafe.Slice((*float64)(unsafe.Pointer(dists)), int(n)) {
		if !s.yield(d) {
			return 0 // early stop (unifrac.go:222-224): the remaining sub-shards are never computed
		}
	}
	return 1
}

// unifracDistsGPU replaces unifracDists (frcfrc/unifrac.go:209-228).  The second result reports
// what went wrong, if anything, once the sequence has been ranged over (the bufio.Scanner
// pattern: unifracDists itself has no error path, a device has); the caller hands it to
// common.ExitIfError, which prints "ERROR: ..." and exits 2 (common/common.go:13-18).
func unifracDistsGPU(nodes [][]flatNode, treeDists []float64, weighted bool) (iter.Seq[float64], func() error) {
	var failure error
	seq := func(yield func(float64) bool) {
		// [][]flatNode (unifrac.go:137-140) -> CSR; lists are sorted by id (unifrac.go:57-59) unless -l (below)
		n := len(nodes)
		indptr := make([]C.int64_t, n+1)
		nnz := 0
		for i, s := range nodes {
			nnz += len(s)
			indptr[i+1] = C.int64_t(nnz)
		}
		ids := make([]C.int32_t, max(nnz, 1))
		abnd := make([]C.double, max(nnz, 1))
		k := 0
		for _, s := range nodes {
			for _, f := range s {
				ids[k], abnd[k] = C.int32_t(f.id), C.double(f.abnd)
				k++
			}
		}
		lens := treeDists
		if len(lens) == 0 {
			lens = make([]float64, 1)
		}
		var o C.ff_options
		C.ff_options_default(&o)
		if weighted {
			o.weighted = 1
		}
		if *nnorm {
			// Under -l the reference hands unifracDists the lists as the recursion left them: normalizeFlatNodes
			// is skipped and with it the sort (unifrac.go:57-59,108-110).  The flag makes the library walk them
			// as they stand, which is what unifracDists does -- a drop-in changes no value, quirks included.
			// (A maintainer who wants the intended -l sorts the lists before this call and drops the flag.)
			o.flags = C.FF_FLAG_UNSORTED_WALK
		}
		errbuf := make([]C.char, 1024)
		h := cgo.NewHandle(&gpuSeq{yield: yield})
		defer h.Delete()
		rc := C.ff_unifrac_dists_stream_csr(C.int64_t(n), C.int64_t(len(treeDists)),
			(*C.double)(unsafe.Pointer(unsafe.SliceData(lens))), unsafe.SliceData(indptr),
			unsafe.SliceData(ids), unsafe.SliceData(abnd), &o, 0 /* 2^25 distances per piece */,
			C.ff_dists_fn(C.ffDeliver), unsafe.Pointer(&h),
			unsafe.SliceData(errbuf), C.size_t(len(errbuf)))
		if rc != 0 {
			failure = fmt.Errorf("%s", C.GoString(unsafe.SliceData(errbuf)))
		}
	}
	return seq, func() error { return failure }
}

// ---- the same with the printing loop included (frcfrc/frcfrc.go:58-62) ----------------------------------------
//
// `for f := range unifrac(...) { fmt.Fprintln(w, f) }` formats 134 M distances of a 16,384-sample run on one
// goroutine: seconds, where the device needs 0.08 s for the distances themselves.  unifracTextGPU hands w the same
// bytes, formatted on the device (same digits: strconv's shortest 'g'), in pieces of whole lines.

// gpuText is what one run shares with its callback.
type gpuText struct {
	w   io.Writer
	err error
}

// ffWriteText receives the next piece of the output: text[0:n] are whole lines (library-owned host memory, valid
// during the call), in order, on the calling goroutine's thread.
//
//export ffWriteText
func ffWriteText(user unsafe.Pointer, text *C.char, n C.size_t) C.int {
	t := (*(*cgo.Handle)(user)).Value().(*gpuText)
	if _, t.err = t.w.Write(unsafe.Slice((*byte)(unsafe.Pointer(text)), int(n))); t.err != nil {
		return 0 // the loop's `break` on a write error (frcfrc.go:60): nothing further is computed
	}
	return 1
}

// flatCSR converts [][]flatNode (unifrac.go:137-140) to the arrays the *_csr entry points take.
func flatCSR(nodes [][]flatNode) (indptr []C.int64_t, ids []C.int32_t, abnd []C.double) {
	nnz := 0
	indptr = make([]C.int64_t, len(nodes)+1)
	for i, s := range nodes {
		nnz += len(s)
		indptr[i+1] = C.int64_t(nnz)
	}
	ids = make([]C.int32_t, max(nnz, 1))
	abnd = make([]C.double, max(nnz, 1))
	k := 0
	for _, s := range nodes {
		for _, f := range s {
			ids[k], abnd[k] = C.int32_t(f.id), C.double(f.abnd)
			k++
		}
	}
	return
}

// unifracTextGPU replaces unifracDists AND the loop that prints its values: it writes to w what that loop writes.
func unifracTextGPU(w io.Writer, nodes [][]flatNode, treeDists []float64, weighted bool) error {
	indptr, ids, abnd := flatCSR(nodes)
	lens := treeDists
	if len(lens) == 0 {
		lens = make([]float64, 1)
	}
	var o C.ff_options
	C.ff_options_default(&o)
	if weighted {
		o.weighted = 1
	}
	if *nnorm {
		o.flags = C.FF_FLAG_UNSORTED_WALK // (as in unifracDistsGPU)
	}
	errbuf := make([]C.char, 1024)
	t := &gpuText{w: w}
	h := cgo.NewHandle(t)
	defer h.Delete()
	rc := C.ff_unifrac_text_stream_csr(C.int64_t(len(nodes)), C.int64_t(len(treeDists)),
		(*C.double)(unsafe.Pointer(unsafe.SliceData(lens))), unsafe.SliceData(indptr),
		unsafe.SliceData(ids), unsafe.SliceData(abnd), &o, 0 /* 2^25 distances per sub-shard */,
		C.ff_text_fn(C.ffWriteText), unsafe.Pointer(&h),
		unsafe.SliceData(errbuf), C.size_t(len(errbuf)))
	if t.err != nil {
		return t.err
	}
	if rc != 0 {
		return fmt.Errorf("%s", C.GoString(unsafe.SliceData(errbuf)))
	}
	return nil
}
