"""Per-kernel ISA statistics (loads, branches, full vmcnt waits, scratch) for a hipcc -S listing.

Usage: hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only x.hip -o x.s && python tools/isa_scan.py x.s
A kernel with about as many `s_waitcnt vmcnt(0)` as global loads is serialising its loads.
"""
import re
import sys

WAIT = re.compile(r"vmcnt\(0\)")
for path in sys.argv[1:]:
    s = open(path).read()
    for m in re.finditer(r"^(_Z\w+):.*?s_endpgm", s, re.S | re.M):
        name, body = m.group(1), m.group(0)
        print("%-70s gload %4d bload %4d branch %4d vmcnt0 %4d scratch %3d mfma %4d" % (
            name[:70], body.count("global_load"), body.count("buffer_load"), body.count("s_cbranch"),
            len(WAIT.findall(body)), body.count("scratch_"), body.count("v_mfma")))
