// rotate-yolov3_amd/csrc/build_id.hip -- the identity of the build: a hash of every translation unit, every header and the compiler
// flags, computed by __graft_entry__.source_id() and passed in as -DRYOLO_BUILD_ID.  bench.py and smoke() print it next to the hash
// of the tree they run in, so a stale library (git-ignored, shipped to the GPU box as a file) cannot pass for HEAD.
#include "../../include/ryolo.h"

#ifndef RYOLO_BUILD_ID
#define RYOLO_BUILD_ID "unstamped0000000"
#endif

extern "C" const char *ryolo_build_id(void) { return "RYOLO_BUILD_ID=" RYOLO_BUILD_ID; }
