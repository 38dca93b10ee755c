timeout 600 python -m pytest tests/test_pack_blocks_gpu.py -q -m gpu -x --tb=short -p no:cacheprovider 2>&1 | tail -12 | cut -c1-220
echo "== pack on"; timeout 300 python tools/pgo_loop.py 100000 400000 4 0 2>&1 | tail -2
echo "== pack off"; timeout 300 python - <<'PY' 2>&1 | tail -2
import sys, runpy
sys.argv = ["tools/pgo_loop.py", "100000", "400000", "4", "0"]
from pypose_amd.optim import posegraph
posegraph.FusedPCG.pack_blocks = False
runpy.run_path("tools/pgo_loop.py", run_name="__main__")
PY
