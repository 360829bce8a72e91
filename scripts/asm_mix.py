"""Static VALU instruction mix of k_lin in the gfx950 asm, priced with the measured issue costs of scripts/microbench (profiles/
rNN_valu_microbench.json): the weighted mean cycles per VALU instruction that bench.py's `roofline_valu_issue` uses.
usage: python scripts/asm_mix.py [kernel-name pattern] [microbench json]  ->  JSON on stdout"""
import collections, glob, json, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pat = sys.argv[1] if len(sys.argv) > 1 else "k_linILi0ELb1ELb1E"
mb_file = sys.argv[2] if len(sys.argv) > 2 else sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_valu_microbench.json")))[-1]
mb = json.load(open(mb_file))["ops"]
asm = "/tmp/dcreg_ctx.s"
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-S", "-o", asm, "--cuda-device-only",
                       os.path.join(ROOT, "dcreg_amd/csrc/device/context.hip")], stderr=subprocess.DEVNULL)
lines = open(asm).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_ZN5dcreg.*%s.*:" % pat, l))
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
ins = [l.strip().split()[0] for l in lines[start + 1:end] if l.startswith("\t") and not l.strip().startswith((".", ";"))]
valu = [i for i in ins if i.startswith("v_") and not i.startswith(("v_mfma", "v_accvgpr"))]


def klass(m):
    if m.startswith(("v_rcp_f64", "v_rsq_f64", "v_sqrt_f64")): return "v_rcp_f64"
    if m.startswith(("v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_exp", "v_log")): return "v_rcp_f32"
    if m.startswith("v_cvt") and "f64" in m: return "v_cvt_f64_f32"
    if "f64" in m: return "v_fma_f64"
    if m.startswith("v_cmp"): return "v_cmp_lt_f32 (e64, sgpr pair)"
    if m.startswith("v_cndmask"): return "v_cndmask_b32"
    if m.startswith("v_pk_"): return "v_pk_mul_f32"
    if m.startswith(("v_add_u32", "v_sub_u32", "v_lshl", "v_lshr", "v_and", "v_or", "v_xor", "v_mov", "v_add_co", "v_addc", "v_mad_u", "v_mul_lo", "v_mul_hi", "v_bfe", "v_ashr", "v_readlane", "v_readfirstlane")): return "v_add_u32"
    return "v_fma_f32"


cnt = collections.Counter(klass(m) for m in valu)
out = {"kernel": lines[start].rstrip(":"), "microbench": os.path.relpath(mb_file, ROOT), "static_valu_instructions": len(valu), "classes": {}}
for w in ("w4", "w8"):
    out["mean_cycles_per_valu_" + w] = sum(cnt[k] * mb[k][w] for k in cnt) / max(len(valu), 1)
for k, v in cnt.most_common():
    out["classes"][k] = {"count": v, "share": v / len(valu), "cycles_w4": mb[k]["w4"], "cycles_w8": mb[k]["w8"]}
print(json.dumps(out, indent=1))
