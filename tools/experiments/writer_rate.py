"""Rate of ff_write_distances (the Go-compatible text formatter) by thread count, into /tmp and into the repo's gpurun_out:
8.4 M values, best of three.
"""
import time, numpy as np, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import frackyfrac_amd as ff
d = np.random.default_rng(1).random(8_386_560)
for path in ("/tmp/ff_out.txt", os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "gpurun_out", "ff_out.txt")):
    for th in (1, 4, 16):
        ts = []
        for rep in range(3):
            t0 = time.perf_counter(); ff.write_distances(path, d, th); ts.append(time.perf_counter() - t0)
        print(path[:12], "threads", th, "best %.3f s  %.1f M values/s" % (min(ts), len(d) / min(ts) / 1e6), flush=True)
    os.remove(path)
print("cpus", len(os.sched_getaffinity(0)))
