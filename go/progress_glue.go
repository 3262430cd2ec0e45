//go:build !js

package main

/*
#include "rt_backend.h"

extern void goProgress(char *json, void *user);   // exported from native_backend.go

void install_progress(rtb_context *c) { rtb_set_progress_callback(c, (rtb_progress_fn)goProgress, NULL); }
*/
import "C"
