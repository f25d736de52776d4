"""Print VGPR / spill / occupancy per kernel from hipcc -Rpass-analysis=kernel-resource-usage."""
import re
import subprocess
import sys

src = sys.argv[1]
r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
                    "-Rpass-analysis=kernel-resource-usage"] + (["-mllvm", "-amdgpu-mfma-vgpr-form=1"] if "attention" in src else []) + ["-c", src, "-o", "/dev/null"], capture_output=True, text=True)
blocks = re.split(r"remark: [^\n]*Function Name: ", r.stderr)[1:]
for b in blocks:
    name = b.split("\n")[0].strip().split()[0]
    dn = subprocess.run(["/usr/bin/c++filt", name], capture_output=True, text=True).stdout.strip()
    dn = re.sub(r"\(.*", "", dn)[:100]

    def g(k):
        m = re.search(k + r": (\d+)", b)
        return int(m.group(1)) if m else -1

    print("%-102s vgpr=%d agpr=%d spill=%d scratch=%d occ=%d lds=%d" % (
        dn, g("VGPRs"), g("AGPRs"), g("VGPR Spill"), g(r"ScratchSize \[bytes/lane\]"),
        g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]")))
