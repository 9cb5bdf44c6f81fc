"""What hipcc made of the MFMA kernels, from `hipcc -S` (no GPU needed): per kernel the spilled registers, the scratch bytes, v_accvgpr moves,
and how many v_mfma write a register quad OTHER than the one they add to.  Round 6 found conv_w2d's "spills" to be exactly that: in
straight-line code hipcc moves accumulators with untied MFMA destinations, both copies live at once (profiles/NOTES.md R6.2b).

    python tools/isa_check.py [csrc/*.hip ...]        default: the instantiation units of the routed MFMA kernels"""
import collections, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "aicovergen_amd", "csrc")
DEFAULT = ["conv_w2d_1.hip", "conv_g1w_1.hip", "conv_g1w_4.hip", "conv_g1_1.hip", "conv_g1_2.hip", "conv_g1s_1.hip", "tdf_pair.hip", "attn.hip"]


def compile_asm(src, extra=()):
    out = tempfile.NamedTemporaryFile(suffix=".s", delete=False).name
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-Wno-unused-result", "-Wno-constant-logical-operand",
           *extra, src, "-o", out]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed: %s\n%s" % (" ".join(cmd), r.stderr[-2000:]))
    text = open(out).read()
    os.remove(out)
    return text


def analyse(text):
    """{kernel symbol: dict(spill, scratch, vgpr, tied, untied, zero, accmov)}"""
    res = collections.OrderedDict()
    name = None
    for line in text.split("\n"):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            name = m.group(1)
            res[name] = dict(spill=None, scratch=None, vgpr=None, tied=0, untied=0, zero=0, accmov=0, scratch_ops=0)
            continue
        if name is None:
            continue
        m = re.match(r"\s*v_mfma_\w+ (\S+), (\S+), (\S+), (\S+)", line)
        if m:
            d, _, _, c = [x.rstrip(",") for x in m.groups()]
            res[name]["zero" if c == "0" else "tied" if c == d else "untied"] += 1
        if "v_accvgpr" in line:
            res[name]["accmov"] += 1
        if re.match(r"\s*scratch_", line):
            res[name]["scratch_ops"] += 1
    # the metadata block: .name / .private_segment_fixed_size / .vgpr_count / .vgpr_spill_count
    cur = None
    for line in text.split("\n"):
        m = re.match(r"\s*\.name:\s+(\S+)", line)
        if m and m.group(1) in res:
            cur = m.group(1)
        if cur:
            for key, field in (("scratch", "private_segment_fixed_size"), ("vgpr", "vgpr_count"), ("spill", "vgpr_spill_count")):
                m = re.match(r"\s*\.%s:\s+(\d+)" % field, line)
                if m:
                    res[cur][key] = int(m.group(1))
    return collections.OrderedDict((k, v) for k, v in res.items() if v["tied"] + v["untied"] + v["zero"])


if __name__ == "__main__":
    srcs = sys.argv[1:] or [os.path.join(CSRC, f) for f in DEFAULT]
    for src in srcs:
        for k, v in analyse(compile_asm(src)).items():
            dem = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip() or k
            print("%-28s %-70s vgpr %3s spill %3s scratch %4s B (%d ops) | mfma tied %3d untied %3d from-zero %3d | accvgpr moves %d"
                  % (os.path.basename(src), dem[:70], v["vgpr"], v["spill"], v["scratch"], v["scratch_ops"], v["tied"], v["untied"], v["zero"], v["accmov"]))
