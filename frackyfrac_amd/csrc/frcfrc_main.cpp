// frcfrc_main.cpp -- the `frcfrc` executable: ff_frcfrc_main and nothing else.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <unistd.h>

#include "frackyfrac_amd.h"

// A profiler's tool library (rocprofv3) writes its output from exit handlers: under one, leave the ordinary way.
static bool under_a_profiler()
{
    const char *pre = getenv("LD_PRELOAD");
    return getenv("ROCP_TOOL_LIBRARIES") || getenv("HSA_TOOLS_LIB") || (pre && strstr(pre, "rocprof"));
}

int main(int argc, char **argv)
{
    const bool fast = !under_a_profiler();
    if (fast) setenv("FF_CLI_FAST_EXIT", "1", 1);  // (ff_frcfrc_main is also a library call: only this process may skip frees)
    const int rc = ff_frcfrc_main(argc, argv);
    if (!fast) return rc;
    // Everything the command owns is closed and flushed by now.  What a plain return would still run is the HIP
    // runtime's own tear-down (queues, code objects, the context of every device: 0.1 s) for a process that is
    // about to give all of it back to the kernel anyway.
    fflush(stdout);
    fflush(stderr);
    _exit(rc);
}
