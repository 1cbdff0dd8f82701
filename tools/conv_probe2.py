import os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
shapes = ["32,96,72,32,32,3,1", "32,48,36,64,64,3,1", "32,24,18,128,128,3,1", "32,12,9,256,256,3,1", "32,96,72,64,256,1,1", "32,96,72,256,64,1,1", "32,96,72,32,64,3,2"]
for s in shapes:
    for f in ("auto", "4", "0", "2"):
        env = dict(os.environ)
        if f != "auto":
            env["STL_CONV_SHAPE"] = f
        r = subprocess.run([sys.executable, "tools/conv_one.py", s, "20"], env=env, capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("B")]
        dbg = [l for l in r.stderr.splitlines() if "[stl conv]" in l]
        print(f"shape={f:4s}", line[0] if line else r.stderr[-300:], "|", dbg[0].split(":")[1].strip() if dbg else "", flush=True)
