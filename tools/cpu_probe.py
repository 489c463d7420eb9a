"""How many host cores does this box really give us?  (sizes the oracle worker pool of tests/conftest.py)
cgroup CPU quota, and the oracle's SuperPoint time at several OpenMP thread counts, alone and two at a time."""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    if os.path.exists(f):
        print(f, open(f).read().strip())
print("affinity", len(os.sched_getaffinity(0)), "loadavg", os.getloadavg())
code = ("import sys,time;sys.path.insert(0,%r);sys.path.insert(0,%r+'/tests');from conftest import load_pkg;U=load_pkg();"
        "from oracle import oracle as O;b=U.synth.pack_sp(U.synth.sp_weights(0));f=U.synth.shift_stream(100,40,480,640)[:3];"
        "O.sp_infer(b,O.SPConfig(1000,0.0005,4),f[0][:64,:64].copy());t=time.time();[O.sp_infer(b,O.SPConfig(1000,0.0005,4),x) for x in f];"
        "print('%%.2f s/frame'%%((time.time()-t)/3))") % (ROOT, ROOT)
for nt in (4, 8, 16, 32, 64):
    for conc in (1, 2, 4):
        env = dict(os.environ, OMP_NUM_THREADS=str(nt))
        t = time.time()
        ps = [subprocess.Popen([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, text=True) for _ in range(conc)]
        outs = [p.communicate()[0].strip() for p in ps]
        print(f"threads {nt:3d} x {conc} concurrent: {outs}  wall {time.time() - t:.1f} s", flush=True)
