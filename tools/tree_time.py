"""Times zkr_balance_tree_build (csrc/rollup_gpu.hip) on a balance tree of 2^20 random leaves: python3 tools/tree_time.py"""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "simple-zk-rollups_amd", "python"))
from zkr_hip.binding import lib  # noqa: E402

d = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 1 << d
buf = bytearray(os.urandom(32 * n))
for i in range(31, len(buf), 32):
    buf[i] &= 0x1F                      # below r
buf = bytes(buf)
out = ctypes.create_string_buffer(32 * (2 * n - 1))
lib().zkr_balance_tree_build(buf[:128], 2, out, 0)
for _ in range(3):
    t = time.time()
    rc = lib().zkr_balance_tree_build(buf, d, out, 0)
    el = time.time() - t
    print("depth %d: rc %d, %.1f ms for %d hashes of two elements (%.1f G Fr-mul/s incl. PCIe both ways)" % (d, rc, 1e3 * el, n - 1, (n - 1) * 440 * 3 / el / 1e9))
