// api.hip -- version / error plumbing of libw2v2hip.so
#include "common.h"

thread_local char g_w2v2_err[512] = {0};

extern "C" int w2v2_version(void) { return 100; }
extern "C" const char* w2v2_last_error(void) { return g_w2v2_err; }
