// unifrac_gpu.go -- the cgo shim that puts libfrackyfrac_amd behind frcfrc's pairwise stage.
//
// Drop this file next to frcfrc/unifrac.go (package main) and have unifrac() (unifrac.go:123)
// return unifracDistsGPU(nodes, treeDists, weighted) instead of unifracDists(...).  It keeps the
// reference's shape -- an iter.Seq[float64] in common.IterPairs order that stops computing when
// the consumer stops (unifrac.go:222-224) -- by staging once and walking the pair space shard by
// shard.  Source only: this image has no Go toolchain.  The call sequence below is exercised,
// call for call, by tests/harness/go_shim_sequence.c on the reference's golden files.
//
// No Go pointer is retained by C after a call returns (cgo rule): every buffer is a Go slice
// that is only borrowed for the duration of the call.
package main

/*
#cgo CFLAGS: -I${SRCDIR}/../../include
#cgo LDFLAGS: -L${SRCDIR}/../../frackyfrac_amd/lib -lfrackyfrac_amd
#include <stdlib.h>
#include "frackyfrac_amd.h"
*/
import "C"

import (
	"fmt"
	"iter"
	"unsafe"
)

// pairsPerShard bounds the host memory of one step: 2^25 distances = 256 MB.
const pairsPerShard = 1 << 25

// unifracDistsGPU replaces unifracDists (frcfrc/unifrac.go:209-228).
func unifracDistsGPU(nodes [][]flatNode, treeDists []float64, weighted bool) (iter.Seq[float64], error) {
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
	for _, s := range nodes { // lists are sorted by id (normalizeFlatNodes, unifrac.go:57-59)
		for _, f := range s {
			ids[k], abnd[k] = C.int32_t(f.id), C.double(f.abnd)
			k++
		}
	}
	var p C.ff_problem
	p.n_samples, p.n_branches = C.int64_t(n), C.int64_t(len(treeDists))
	p.branch_len = (*C.double)(unsafe.Pointer(unsafe.SliceData(treeDists)))
	p.indptr = unsafe.SliceData(indptr)
	p.branch_id = unsafe.SliceData(ids)
	p.abnd = unsafe.SliceData(abnd)

	var o C.ff_options
	C.ff_options_default(&o)
	if weighted {
		o.weighted = 1
	}
	errbuf := make([]C.char, 1024)
	eb, el := unsafe.SliceData(errbuf), C.size_t(len(errbuf))
	fail := func() error { return fmt.Errorf("%s", C.GoString(eb)) } // common.ExitIfError prints "ERROR: ..."

	var plan *C.ff_plan
	if C.ff_plan_create(&p, &o, &plan, eb, el) != 0 { // flat nodes -> HBM, staged once
		return nil, fail()
	}
	shards := C.int32_t(C.ff_num_pairs(C.int64_t(n))/pairsPerShard + 1)
	return func(yield func(float64) bool) {
		defer func() { C.ff_plan_destroy(plan) }()
		for r := C.int32_t(0); r < shards; r++ {
			if C.ff_plan_set_shard(plan, r, shards, eb, el) != 0 {
				panic(fail())
			}
			var info C.ff_plan_info
			C.ff_plan_info_get(plan, &info)
			m := int(info.slot_end - info.slot_begin)
			if m == 0 {
				continue
			}
			part := make([]float64, m)
			rc := C.ff_plan_run_host(plan, (*C.double)(unsafe.Pointer(unsafe.SliceData(part))), eb, el)
			if rc == C.FF_ERR_PRECISION { // a data set of replicates: binary64 from here on
				C.ff_plan_destroy(plan)
				plan = nil
				o.precision = C.FF_PRECISION_EXACT64
				if C.ff_plan_create(&p, &o, &plan, eb, el) != 0 ||
					C.ff_plan_set_shard(plan, r, shards, eb, el) != 0 {
					panic(fail())
				}
				rc = C.ff_plan_run_host(plan, (*C.double)(unsafe.Pointer(unsafe.SliceData(part))), eb, el)
			}
			if rc != 0 {
				panic(fail())
			}
			for _, d := range part { // slots slot_begin .. slot_end-1 of common.IterPairs order
				if !yield(d) {
					return // early stop: the remaining shards are never computed
				}
			}
		}
	}, nil
}
