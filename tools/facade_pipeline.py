"""The facade_pipeline leg of bench.py alone (witness -> proof -> verify over 256 batches of the tx circuit):
python tools/facade_pipeline.py [n_batches] [chunk]"""
import json, os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "simple-zk-rollups_amd", "python"))
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
c = int(sys.argv[2]) if len(sys.argv) > 2 else 64
w = sys.argv[3] if len(sys.argv) > 3 else "gpu"
print(json.dumps(bench.facade_pipeline_leg(0, n, c, w)))
