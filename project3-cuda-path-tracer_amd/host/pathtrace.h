// pathtrace.h -- the reference's renderer API (src/pathtrace.h:6-8), provided by pathtrace_shim.cpp
// on top of the C ABI in include/pt_amd.h.
#pragma once
#include "scene.h"

#ifndef PT_HAVE_UCHAR4
struct uchar4 { unsigned char x, y, z, w; };   // the host side needs no HIP header for this
#endif

void pathtraceInit(Scene *scene);
void pathtraceFree();
// `pbo` is a DEVICE pointer to W*H uchar4 (the mapped GL buffer in the reference) or NULL when headless.
void pathtrace(uchar4 *pbo, int frame, int iteration);
