// pathtrace.h -- the reference's renderer API (its src/pathtrace.h:6-8), provided by pathtrace_shim.cpp on top of the
// C ABI in include/pt_amd.h.  Same three names, argument meaning and error behaviour (message + exit(EXIT_FAILURE)).
#pragma once
#include "scene.h"

#ifndef PT_HAVE_UCHAR4
struct uchar4 { unsigned char x, y, z, w; };   // 4-byte RGBA texel; the host side needs no GPU header for it
#endif

// Uploads the scene and allocates the device state.  `scene` is borrowed until pathtraceFree(); call it again
// (after pathtraceFree) whenever the camera changed -- the accumulator restarts from zero.
void pathtraceInit(Scene *scene);

// Releases the device state.  Legal before the first pathtraceInit (the reference's driver calls it first).
void pathtraceFree();

// One iteration = 1 sample per pixel; `iteration` is 1-based and increasing, `frame` is always 0.
// `pbo`: DEVICE pointer to W*H uchar4 (the mapped GL buffer in the reference) or NULL when headless.
// On return scene->state.image holds the un-normalised running sum (divide by `iteration` to display).
void pathtrace(uchar4 *pbo, int frame, int iteration);
