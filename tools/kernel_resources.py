"""Register / scratch / occupancy table of one HIP source (hipcc -Rpass-analysis=kernel-resource-usage).
usage: python tools/kernel_resources.py csrc/attention.hip [filter-substring] [extra hipcc flags...]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "group-attribution-for-diffusion-models_amd")
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
extra = sys.argv[3:]
if "attention" in src:
    extra += ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", f"-I{ROOT}/include", f"-I{PKG}/csrc",
       "-Wno-unused-result", "-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(PKG, src), "-o", "/tmp/_kr.o"] + extra
txt = subprocess.run(cmd, capture_output=True, text=True).stderr
for b in re.split(r"remark: [^\n]*Function Name: ", txt)[1:]:
    name = b.split("\n")[0].strip()
    def g(k):
        m = re.search(k + r": (\d+)", b)
        return int(m.group(1)) if m else -1
    try:
        name = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    except FileNotFoundError:
        pass
    name = name.replace("(anonymous namespace)::", "").replace("((anonymous namespace)::AttnDev)", "").replace("((anonymous namespace)::DevArgs)", "")
    if flt in name:
        scratch, occ, lds = g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]")
        print(f"{name:80s} vgpr {g('VGPRs'):4d} agpr {g('AGPRs'):4d} scratch {scratch:5d} occ {occ} lds {lds}")
