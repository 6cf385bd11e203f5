"""One-off: the C3 init product with the roles of the two operands swapped (C' = B'A'), to see
how the workgroup -> XCD mapping of the tile grid changes the rate."""
import sys
sys.argv = sys.argv[:1]
import importlib.util, types
src = open("scripts/perf_gemm.py").read().split('run("mat_x_m1')[0]
ns = {}
exec(compile(src, "perf_gemm_head", "exec"), ns)
ns["run"]("C3 init, as is", 1000, 10000, 50000, False, False)
ns["run"]("C3 init, operands swapped", 10000, 1000, 50000, False, False)
