#!/usr/bin/env python3
"""Every kernel of liblavt_hip.so with VGPR spills or scratch: `llvm-readelf --notes` (AMDGPU metadata) of the gfx950 code objects bundled in the library.
usage: spill_table.py [liblavt_hip.so] > profiles/rNN_vgpr_spills_readelf.txt      (runs without a GPU)"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "lavt-rs_amd", "csrc", "liblavt_hip.so")
LLVM = "/opt/rocm/lib/llvm/bin"
tmp = tempfile.mkdtemp()
work = os.path.join(tmp, "lib.so")
subprocess.check_call(["cp", lib, work])
subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", work], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
kernels = []
for f in sorted(os.listdir(tmp)):
    if "gfx950" not in f:
        continue
    notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", os.path.join(tmp, f)], capture_output=True, text=True).stdout
    for blk in re.split(r"\n\s+- \.agpr_count:", notes)[1:]:
        get = lambda key: (re.search(r"\." + key + r":\s+(.+)", blk) or [None, "?"])[1].strip()
        name = get("name")
        try:
            demangled = subprocess.run(["c++filt", name.strip("'")], capture_output=True, text=True).stdout.strip() or name
        except Exception:
            demangled = name
        kernels.append((int(get("vgpr_count")), int(get("vgpr_spill_count")), int(get("private_segment_fixed_size")), demangled))
bad = [k for k in kernels if k[1] or k[2]]
print(f"{len(kernels)} kernels in liblavt_hip.so (gfx950); {len(bad)} with VGPR spills or scratch  (llvm-readelf --notes of the extracted code objects; tools/spill_table.py)")
print("vgpr spill scratch_bytes  kernel")
for v, s, sc, n in sorted(bad, key=lambda k: -k[1]):
    print(f"{v:4d} {s:5d} {sc:7d}  {n[:200]}")
