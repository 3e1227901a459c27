#!/usr/bin/env python3
"""Writes tests/golden/circuit_data_gadget_and.bin: `CircuitData::to_bytes` (as restated in circuit_bytes.cpp) of the
and(x, y) gadget circuit, produced on the GPU box (the constants/sigmas commitment in it comes from the device).
The CPU tests parse it (reader, validation, fuzz) -- it pins this library's writer against its reader across rounds,
not against upstream (no upstream artifact of this form exists in the reference)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
p25 = ge.load_package(); p25.device_init(0)
c = p25.Circuit.build_gadget(0, 0)
data = c.to_bytes()
out = os.path.join(ROOT, "gpurun_out", "circuit_data_gadget_and.bin")
open(out, "wb").write(data)
print(len(data), "bytes ->", out, " degree_bits", int(c.info.degree_bits))
