"""VGPRs / spills / scratch / occupancy of every kernel in kernels.hip, from hipcc's -Rpass-analysis=kernel-resource-usage remarks
(no GPU needed).  usage: python3 scripts/kernel_resource_usage.py > profiles/rNN_kernel_resource_usage.txt"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "fwumious_wabbit_amd", "csrc")
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "--offload-arch=gfx950", f"-I{ROOT}/include", "-I.", "-mllvm", "-pragma-unroll-threshold=131072",
       "-Rpass-analysis=kernel-resource-usage", "-c", "kernels.hip", "-o", "/tmp/kernels_ru.o"]
out = subprocess.run(cmd, cwd=src, capture_output=True, text=True).stderr
# (kernels.hip is two translation units: the phase kernels -- FWD / UPD / MID -- are the second, see the Makefile)
out += subprocess.run(cmd[:-4] + ["-DFW_PHASE_TU"] + cmd[-4:], cwd=src, capture_output=True, text=True).stderr
rows, cur = [], None
for line in out.splitlines():
    m = re.search(r"remark:\s+(.*?)(?: \[-Rpass)", line)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = {"name": t.split(":", 1)[1].strip()}
        rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
dem = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), capture_output=True, text=True).stdout.splitlines()
print("# hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -Rpass-analysis=kernel-resource-usage -c kernels.hip, then the same with -DFW_PHASE_TU   (scripts/kernel_resource_usage.py)")
print("# template arguments: fw_example_kernel_r<OPT (100 SGD / 200 AdagradFlex / 300 AdagradLUT), COH (updating launch), MAXR, WIN (chained update path), NC (16-byte chunks per row)>")
print("#                     fw_example_kernel<VEC, OPT, COH, PH (0 fused / 1 FWD / 3 UPD), NN (deep head)>")
print("# kernel, VGPRs, AGPRs, scratch bytes/lane, occupancy waves/SIMD, SGPR spills (to VGPR lanes), VGPR spills, static LDS")
for r, d in zip(rows, dem):
    d = d.replace("void fwgpu::", "").replace("(fwgpu::KernelParams)", "")
    print(", ".join([d, r.get("VGPRs", "?"), r.get("AGPRs", "?"), r.get("ScratchSize [bytes/lane]", "?"), r.get("Occupancy [waves/SIMD]", "?"),
                     r.get("SGPRs Spill", "?"), r.get("VGPRs Spill", "?"), r.get("LDS Size [bytes/block]", "?")]))
