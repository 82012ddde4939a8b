"""What the GPU box offers an out-of-core factorisation: host RAM and the cgroup's share of it (READ from /proc and
/sys, nothing is allocated to find out), pinned H2D / D2H bandwidth on a small fixed buffer.

Round 5 lesson: a first version of this probe pinned host memory in 16 GiB steps "until it stops" and took the box
down with it.  Never size host memory by trying."""
import os
import sys
import time


def read(p):
    try:
        return open(p).read().strip()
    except OSError as e:
        return f"<{e.__class__.__name__}>"


def main():
    mi = {l.split(":")[0]: l.split(":")[1].strip() for l in read("/proc/meminfo").splitlines() if ":" in l}
    for k in ("MemTotal", "MemFree", "MemAvailable", "SwapTotal", "Mlocked", "Unevictable", "HugePages_Total"):
        print(k, mi.get(k))
    for p in ("/sys/fs/cgroup/memory.max", "/sys/fs/cgroup/memory.high", "/sys/fs/cgroup/memory.current",
              "/sys/fs/cgroup/memory/memory.limit_in_bytes", "/sys/fs/cgroup/memory/memory.usage_in_bytes",
              "/proc/self/cgroup", "/sys/fs/cgroup/cpu.max"):
        print(p, read(p).replace("\n", " | "))
    import resource
    print("RLIMIT_MEMLOCK", resource.getrlimit(resource.RLIMIT_MEMLOCK), "RLIMIT_AS", resource.getrlimit(resource.RLIMIT_AS))
    print("cpus", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
    print(os.popen("df -h /dev/shm /tmp 2>&1 | tail -2").read())
    print(os.popen("numactl -H 2>&1 | head -12").read())
    if len(sys.argv) > 1 and sys.argv[1] == "bw":
        import torch
        free, total = torch.cuda.mem_get_info()
        print(f"HBM free {free/2**30:.1f} GiB of {total/2**30:.1f}")
        n = 1 * 2**30                      # 1 GiB, fixed
        h = torch.empty(n, dtype=torch.uint8, pin_memory=True)
        h2 = torch.empty(n, dtype=torch.uint8, pin_memory=True)
        h.fill_(1)
        d = torch.empty(n, dtype=torch.uint8, device="cuda")
        d2 = torch.empty(n, dtype=torch.uint8, device="cuda")
        s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
        for name, fn in (("H2D", lambda: d.copy_(h, non_blocking=True)), ("D2H", lambda: h.copy_(d, non_blocking=True))):
            fn(); torch.cuda.synchronize()
            t0 = time.time()
            for _ in range(5):
                fn()
            torch.cuda.synchronize()
            print(f"{name} pinned: {5*n/(time.time()-t0)/1e9:.1f} GB/s")
        t0 = time.time()
        for _ in range(5):
            with torch.cuda.stream(s1):
                d.copy_(h, non_blocking=True)
            with torch.cuda.stream(s2):
                h2.copy_(d2, non_blocking=True)
        torch.cuda.synchronize()
        print(f"H2D + D2H at once: {10*n/(time.time()-t0)/1e9:.1f} GB/s in all")


if __name__ == "__main__":
    main()
