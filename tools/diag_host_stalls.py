"""Which part of a step's HOST time holds the sporadic 5-55 ms stall of the un-synchronised step loops :
the C-ABI calls (msgs_forward / msgs_backward, i.e. hipLaunchKernel and friends) or the torch side (allocator, autograd)?
Wraps the two entry points with a host timer and reads the caching allocator's hipMalloc / hipFree counters per step.
python tools/diag_host_stalls.py [reps]"""
import gc, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ms-gs_amd"), os.path.join(ROOT, "ms-gs_amd", "host"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch, scenes
import diff_gaussian_rasterization as dgr
from gaussian_renderer import PIPE, render
from synthetic_model import SyntheticGaussians

# a sampler PROCESS (started as a child, before this one touches the GPU) records every ~0.5 ms the scheduler state and
# kernel wait channel of each of this process's threads; time.perf_counter is CLOCK_MONOTONIC, comparable across processes
import subprocess, tempfile
_SAMPLER = r"""
import os, sys, time
pid = int(sys.argv[1]); out = open(sys.argv[2], "w")
while os.path.exists(f"/proc/{pid}"):
    t = time.perf_counter(); row = []
    try:
        for tid in os.listdir(f"/proc/{pid}/task"):
            try:
                st = open(f"/proc/{pid}/task/{tid}/stat").read().rsplit(")", 1)[1].split()
                wc = open(f"/proc/{pid}/task/{tid}/wchan").read().strip()
                sy = open(f"/proc/{pid}/task/{tid}/syscall").read().split()[0]
                row.append(f"{tid}:{st[0]}:{wc}:{sy}:{st[36] if len(st) > 36 else '?'}")
            except OSError:
                pass
    except OSError:
        break
    out.write(f"{t:.6f} " + " ".join(row) + "\n"); out.flush()
    time.sleep(0.0004)
"""
_samp_path = os.path.join(tempfile.gettempdir(), f"msgs_sampler_{os.getpid()}.log")
_samp = subprocess.Popen([sys.executable, "-c", _SAMPLER, str(os.getpid()), _samp_path])
def cpu_stat():
    d = {}
    for f in ("/sys/fs/cgroup/cpu.stat", "/sys/fs/cgroup/cpu/cpu.stat"):
        try:
            for ln in open(f):
                k, v = ln.split(); d[k] = int(v)
            break
        except OSError:
            pass
    try:
        r, w, n = open("/proc/thread-self/schedstat").read().split(); d["run_ns"], d["rq_wait_ns"], d["slices"] = int(r), int(w), int(n)
    except OSError:
        pass
    return d
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try:
        print(f, open(f).read().strip())
    except OSError:
        pass
print("cpus", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "main tid", os.getpid())

lib = dgr._C.lib
spent = {"msgs_forward": 0.0, "msgs_backward": 0.0}
def _wrap(name):
    real = getattr(lib, name)
    def timed(*a):
        t = time.perf_counter()
        r = real(*a)
        spent[name] += time.perf_counter() - t
        return r
    setattr(lib, name, timed)
_wrap("msgs_forward"); _wrap("msgs_backward")

marks = {}
def _wrap_py(owner, name, key, static=False):
    real = getattr(owner, name)
    def timed(*a, **k):
        t = time.perf_counter()
        r = real(*a, **k)
        marks.setdefault(key, []).append((t, time.perf_counter()))
        return r
    setattr(owner, name, staticmethod(timed) if static else timed)
_wrap_py(dgr._RasterizeGaussiansRaw, "backward", "op.backward", static=True)
_wrap_py(dgr, "_check_saved", "check_saved")
_wrap_py(dgr, "_grad_out", "grad_out")
_wrap_py(dgr, "_take_backward_scratch", "scratch")

gc.collect(); gc.disable()
if os.environ.get("DIAG_SINGLE_THREAD_BACKWARD") == "1":
    torch.autograd.set_multithreading_enabled(False)     # backward on the calling thread (no hand-off to the device thread)
sc, cam, st = scenes.config("C3")
pc = SyntheticGaussians(sc, "cuda", requires_grad=True)
bg = torch.zeros(3, device="cuda")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
def mallocs():
    s = torch.cuda.memory_stats()
    return s.get("num_device_alloc", 0), s.get("num_device_free", 0), s.get("num_alloc_retries", 0)
n_steps = n_hic = 0
stalls = []
for rep in range(reps):
    for k in (0, 2, 3, 4):
        W, H = int(1920 / 2 ** k), int(1080 / 2 ** k)
        c = scenes.front_camera(W, H).to("cuda"); dL = scenes.grad_seed(W, H, 40 + k).to("cuda")
        torch.cuda.synchronize()
        for it in range(25):
            marks.clear(); c0 = cpu_stat()
            m0 = mallocs(); spent["msgs_forward"] = spent["msgs_backward"] = 0.0
            t0 = time.perf_counter()
            for p_ in pc.parameters(): p_.grad = None
            out = render(c, pc, PIPE, bg, **st); t1 = time.perf_counter()
            out["render"].backward(dL); t2 = time.perf_counter()
            m1 = mallocs(); n_steps += 1
            if rep + it > 0 and (t2 - t0) > 3e-3:
                n_hic += 1
                print(f"rep {rep} k={k} step {it}: fwd host {1e3*(t1-t0):.2f} ms (msgs_forward {1e3*spent['msgs_forward']:.2f}) "
                      f"bwd host {1e3*(t2-t1):.2f} ms (msgs_backward {1e3*spent['msgs_backward']:.2f}); hipMalloc +{m1[0]-m0[0]} "
                      f"hipFree +{m1[1]-m0[1]} retries +{m1[2]-m0[2]}", flush=True)
                c1 = cpu_stat()
                print("      cgroup/sched deltas: " + " ".join(f"{k}+{c1[k]-c0[k]}" for k in c1 if c1[k] != c0.get(k, 0)))
                stalls.append((t0, t2))
                ob = marks.get("op.backward", [(t1, t1)])[0]
                print(f"      backward(): engine->op {1e3*(ob[0]-t1):.2f} ms, inside op {1e3*(ob[1]-ob[0]):.2f} ms "
                      f"(check_saved {1e3*sum(b-a for a,b in marks.get('check_saved',[])):.2f}, grad_out "
                      + " ".join(f"{1e3*(b-a):.2f}" for a, b in marks.get('grad_out', []))
                      + f", scratch {1e3*sum(b-a for a,b in marks.get('scratch',[])):.2f}), op->return {1e3*(t2-ob[1]):.2f} ms", flush=True)
        torch.cuda.synchronize()
print(f"{n_hic} stalled steps of {n_steps}")

_samp.terminate(); _samp.wait()
# who used the CPU: per-thread user+system time of this process by thread name
tck = os.sysconf("SC_CLK_TCK"); by = {}
for tid in os.listdir("/proc/self/task"):
    try:
        st = open(f"/proc/self/task/{tid}/stat").read()
        name = st[st.index("(") + 1: st.rindex(")")]; f = st.rsplit(")", 1)[1].split()
        e = by.setdefault(name, [0, 0.0]); e[0] += 1; e[1] += (int(f[11]) + int(f[12])) / tck
    except (OSError, ValueError):
        pass
print("threads:", torch.get_num_threads(), "intra-op;", " | ".join(f"{k} x{v[0]} {v[1]:.2f}s" for k, v in sorted(by.items(), key=lambda kv: -kv[1][1])[:8]))
# what the threads were doing during each stalled step
rows = [ln.split() for ln in open(_samp_path)]
for a, b in stalls[-6:]:
    print(f"--- samples inside the stalled step [{a:.4f}, {b:.4f}] (tid:state:wchan:syscall:cpu)")
    inside = [r for r in rows if a <= float(r[0]) <= b]
    for r in inside[:: max(1, len(inside) // 12)]:
        print("   ", f"+{1e3 * (float(r[0]) - a):.2f} ms", " ".join(x for x in r[1:] if not x.split(":")[1] == "S" or x.startswith(str(os.getpid()) + ":")))
os.unlink(_samp_path)
