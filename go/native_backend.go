//go:build !js

// native_backend.go -- the six entry points of the reference backend
// (src/backend/main.go:28-33) as plain Go functions over libraytracer_amd.so
// (cgo, include/rt_backend.h).  NOT BUILT in the environment this repository
// was developed in (it has no Go toolchain); the same ABI is exercised there
// through ctypes and a C99 caller.  Build: `cd go && go build ./...` with the
// library built in ../raytracer_amd/lib.
package main

/*
#cgo CFLAGS: -I${SRCDIR}/../include
#cgo LDFLAGS: -L${SRCDIR}/../raytracer_amd/lib -lraytracer_amd -Wl,-rpath,${SRCDIR}/../raytracer_amd/lib
#include <stdlib.h>
#include "rt_backend.h"

// A file that uses //export may only DECLARE C functions in this preamble;
// install_progress is defined in progress_glue.go.
void install_progress(rtb_context *c);
*/
import "C"
import (
	"errors"
	"fmt"
	"os"
	"unsafe"
)

var ctx *C.rtb_context               // replaces `var context *models.RenderContext` (main.go:20)
var ProgressUpdate func(json string) // replaces js.Global().Call("progressUpdate", …) (utility.go:31)

//export goProgress
func goProgress(json *C.char, _ unsafe.Pointer) {
	if ProgressUpdate != nil {
		ProgressUpdate(C.GoString(json))
	}
}

func lastError() error { return errors.New(C.GoString(C.rt_last_error())) }

func take(p *C.char, n C.size_t) string { // result strings are malloc'd by the library
	defer C.rt_free(unsafe.Pointer(p))
	return C.GoStringN(p, C.int(n))
}

// main.go:39-63
func Initialize(ctxJSON string, textures ...[]byte) error {
	if ctx == nil {
		if C.rtb_create(0, &ctx) != C.RT_OK { // one context per worker / per GPU
			return lastError()
		}
		C.install_progress(ctx)
	}
	cs := C.CString(ctxJSON)
	defer C.free(unsafe.Pointer(cs))
	ptrs := make([]*C.uint8_t, len(textures)+1)
	sizes := make([]C.size_t, len(textures)+1)
	for i, t := range textures {
		ptrs[i] = (*C.uint8_t)(C.CBytes(t))
		defer C.free(unsafe.Pointer(ptrs[i]))
		sizes[i] = C.size_t(len(t))
	}
	if C.rtb_initialize(ctx, cs, &ptrs[0], &sizes[0], C.int32_t(len(textures))) != C.RT_OK {
		return lastError()
	}
	return nil
}

// main.go:65-76
func BuildBVH() (string, error) {
	var p *C.char
	var n C.size_t
	if C.rtb_build_bvh(ctx, &p, &n) != C.RT_OK {
		return "", lastError()
	}
	return take(p, n), nil
}

// main.go:78-90
func LoadBVH(bvhJSON string) error {
	cs := C.CString(bvhJSON)
	defer C.free(unsafe.Pointer(cs))
	if C.rtb_load_bvh(ctx, cs, C.size_t(len(bvhJSON))) != C.RT_OK {
		return lastError()
	}
	return nil
}

// main.go:93-188 — like the reference, errors come back INSIDE the result
// document (exitCode -1, message, blank 500x500 image, main.go:308-325)
func Render(passJSON string) string {
	cs := C.CString(passJSON)
	defer C.free(unsafe.Pointer(cs))
	var p *C.char
	var n C.size_t
	C.rtb_render(ctx, cs, &p, &n)
	return take(p, n)
}

// main.go:196-231, 234-306
func InitializeIncrementalRender(passJSON string) string {
	cs := C.CString(passJSON)
	defer C.free(unsafe.Pointer(cs))
	var p *C.char
	var n C.size_t
	C.rtb_initialize_incremental_render(ctx, cs, &p, &n)
	if p == nil {
		return "" // the reference returns nil on success
	}
	return take(p, n)
}

func IncrementalRender() string {
	var p *C.char
	var n C.size_t
	C.rtb_incremental_render(ctx, &p, &n)
	return take(p, n)
}

// A minimal command-line driver, so that the package builds as a program:
// native_backend <context.json> <pass.json> renders one pass and prints the
// result document (what go_webworker.js:105-120 does with the wasm build).
func main() {
	if len(os.Args) != 3 {
		fmt.Fprintln(os.Stderr, "usage: native_backend <context.json> <pass.json>")
		os.Exit(2)
	}
	ctxJSON, err := os.ReadFile(os.Args[1])
	if err != nil {
		fmt.Fprintln(os.Stderr, err)
		os.Exit(1)
	}
	passJSON, err := os.ReadFile(os.Args[2])
	if err != nil {
		fmt.Fprintln(os.Stderr, err)
		os.Exit(1)
	}
	ProgressUpdate = func(json string) { fmt.Fprintln(os.Stderr, json) }
	if err := Initialize(string(ctxJSON)); err != nil {
		fmt.Fprintln(os.Stderr, err)
		os.Exit(1)
	}
	bvh, err := BuildBVH()
	if err != nil {
		fmt.Fprintln(os.Stderr, err)
		os.Exit(1)
	}
	if err := LoadBVH(bvh); err != nil {
		fmt.Fprintln(os.Stderr, err)
		os.Exit(1)
	}
	fmt.Println(Render(string(passJSON)))
}
