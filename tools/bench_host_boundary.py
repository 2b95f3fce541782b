#!/usr/bin/env python3
"""PCIe-inclusive rates of the drop-in boundary (host pointers in, host arrays out), measured by the C++ caller
tools/hostbench.cpp through the C ABI.  bench.py embeds the same line as its `pcie_inclusive` object; this wrapper
only exists to run it alone:  python tools/bench_host_boundary.py [rows cols batch nfeatures]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


if __name__ == "__main__":
    a = [int(v) for v in sys.argv[1:5]] + [480, 752, 64, 1000][len(sys.argv[1:5]):]
    print(bench.pcie_inclusive(a[0], a[1], a[2], a[3], as_text=True))
