"""cProfile of the host side of bench.py's training step (where the enqueue time goes)."""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = ["bench.py", "--steps", "12", "--warmup", "6", "--no-cpu-baseline", "--no-extras"] + sys.argv[1:]
os.environ.setdefault("PCB_BENCH_NO_ROOFLINE", "1")
import bench  # noqa: E402

pr = cProfile.Profile()
pr.enable()
bench.main()
pr.disable()
st = pstats.Stats(pr, stream=sys.stderr)
st.sort_stats("tottime").print_stats(45)
st.sort_stats("cumulative").print_stats(60)
