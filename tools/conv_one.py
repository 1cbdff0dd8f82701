import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.conv_probe import run
shape = tuple(int(v) for v in sys.argv[1].split(","))
run(*shape, reps=int(sys.argv[2]) if len(sys.argv) > 2 else 5)
