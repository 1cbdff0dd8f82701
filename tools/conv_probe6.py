import os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
shapes = ["32,96,72,32,32,3,1", "32,48,36,64,64,3,1", "32,24,18,128,128,3,1", "32,12,9,256,256,3,1"]
for s in shapes:
    for f in ("1", "2", "4", "7"):
        for cap in ("256", "512", "1024", "2048"):
            env = dict(os.environ, STL_CONV_SHAPE=f, STL_CONV_GRID_CAP=cap)
            env["STL_CONV_WS"] = "1" if f == "7" else "0"
            r = subprocess.run([sys.executable, "tools/conv_one.py", s, "30"], env=env, capture_output=True, text=True)
            line = [l for l in r.stdout.splitlines() if l.startswith("B")]
            print(f"shape={f} cap={cap:5s}", line[0][:95] if line else r.stderr[-200:].replace("\n", " "), flush=True)
