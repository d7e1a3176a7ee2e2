"""Register / LDS / spill table of the kernels in a HIP source (hipcc -Rpass-analysis=kernel-resource-usage).
    python tools/kernel_resources.py stainx_amd/csrc/macenko.hip [name-filter]"""
import re, subprocess, sys
src = sys.argv[1]; flt = sys.argv[2] if len(sys.argv) > 2 else ""
out = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-mllvm", "-amdgpu-mfma-vgpr-form=1", "-c", src, "-o", "/tmp/_res.o", "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True).stderr
blocks = re.split(r"remark: Function Name: ", out)[1:]
keys = [("sgpr", r"TotalSGPRs: (\d+)"), ("vgpr", r" VGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"), ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("s-spill", r"SGPRs Spill: (\d+)"), ("v-spill", r"VGPRs Spill: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)")]
for b in blocks:
    mangled = b.split()[0]
    name = subprocess.run(["c++filt", mangled], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"\(.*", "", name).replace("void sx::macenko::", "")
    if flt and flt not in name: continue
    print(f"{name:60s} " + " ".join(f"{k} {re.search(p, b).group(1):>5}" for k, p in keys))
