"""Developer tool: per-function instruction / register statistics of the gfx950 code object."""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "lunar_module_ascent_trajectory_optimiser_amd", "csrc", os.environ.get("ASM_SRC", "ascent_solver.hip"))
d = tempfile.mkdtemp()
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-I", os.path.join(ROOT, "include"),
                "-save-temps", "-c", "-o", "x.o", src] + sys.argv[1:], cwd=d, check=True, stderr=subprocess.DEVNULL)
s = open(os.path.join(d, [f for f in os.listdir(d) if f.endswith("gfx950.s")][0])).read()
if os.environ.get("KEEP_ASM"):
    open(os.environ["KEEP_ASM"], "w").write(s)
for f in re.split(r"\n(?=_Z[\w]+:)", s):
    m = re.match(r"(_Z\w+):", f)
    if not m:
        continue
    c = lambda pat: len(re.findall(pat, f))
    g = lambda pat: (re.search(pat, f) or [None, "?"])[1]
    P = dict(f64=r"v_(fma|mul|add)_f64", valu=r"\n\s+v_", salu=r"\n\s+s_", gld="global_load", gst="global_store",
             flat="flat_(load|store)", scr="scratch_(load|store)", acc="v_accvgpr", wait="s_waitcnt")
    R = dict(vgpr=r"; NumVgprs: (\d+)", agpr=r"; NumAgprs: (\d+)", scratch=r"; ScratchSize: (\d+)")
    print(m.group(1)[14:60].ljust(46), " ".join(f"{k} {c(v):5d}" for k, v in P.items()), " ".join(f"{k} {g(v)}" for k, v in R.items()))
