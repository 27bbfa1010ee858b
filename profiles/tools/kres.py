#!/usr/bin/env python3
"""Kernel resource usage (VGPR / AGPR / SGPR, spills, scratch) of every kernel in the built library:
    python3 profiles/tools/kres.py [nerf_fl_amd/libnerf_fl_amd.so]
Reads the AMDGPU metadata notes of the gfx950 code object embedded in the .so (no GPU needed)."""
import re
import subprocess
import sys
import tempfile

so = sys.argv[1] if len(sys.argv) > 1 else "nerf_fl_amd/libnerf_fl_amd.so"
data = open(so, "rb").read()
# the fat binary holds one ELF per offload target: take every embedded ELF that is an AMDGPU code object
out = []
for m in re.finditer(b"\x7fELF\x02\x01\x01", data):
    if m.start() == 0:
        continue
    with tempfile.NamedTemporaryFile(suffix=".co") as f:
        f.write(data[m.start():])
        f.flush()
        r = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", f.name], capture_output=True, text=True)
    if "amdhsa.kernels" in r.stdout:
        out.append(r.stdout)
for txt in out:
    for blk in txt.split("- .agpr_count:")[1:]:
        def g(k):
            mm = re.search(r"\." + k + r":\s+(\S+)", blk)
            return mm.group(1) if mm else "?"
        name = subprocess.run(["c++filt", g("name")], capture_output=True, text=True).stdout.strip()
        name = re.sub(r"\(.*", "", name.replace("void ", ""))
        print(f"{name[:60]:60s} vgpr {g('vgpr_count'):>4} agpr {blk.split()[0]:>4} sgpr {g('sgpr_count'):>4} "
              f"sspill {g('sgpr_spill_count'):>4} vspill {g('vgpr_spill_count'):>4} scratch {g('private_segment_fixed_size'):>5} "
              f"lds {g('group_segment_fixed_size'):>6}")
