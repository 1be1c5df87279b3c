// frcfrc_main.cpp -- the `frcfrc` executable: ff_frcfrc_main and nothing else.
#include <cstdio>
#include <unistd.h>

#include "frackyfrac_amd.h"

int main(int argc, char **argv)
{
    const int rc = ff_frcfrc_main(argc, argv);
    // Everything the command owns is closed and flushed by now.  What a plain return would still run is the HIP
    // runtime's own tear-down (queues, code objects, the context of every device: 0.1-0.2 s) for a process that is
    // about to give all of it back to the kernel anyway.
    fflush(stdout);
    fflush(stderr);
    _exit(rc);
}
