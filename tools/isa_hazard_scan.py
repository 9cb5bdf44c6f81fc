"""conv_w2d's MFMAs are inline asm, which hipcc's hazard recogniser does not look into (tools/isa_check.py, profiles/NOTES.md R6.2b).  This
scans the compiled kernels for the two hazards that are therefore handled by hand: (1) a non-MFMA instruction reading a register an asm
MFMA wrote fewer than 11 wait states earlier (an 8-pass MFMA's result), (2) an asm MFMA reading, as its A or B operand, a register a VALU
instruction wrote fewer than 2 wait states earlier.  Wait states are counted the way LLVM counts them: one per instruction, n + 1 for s_nop n.
    python tools/isa_hazard_scan.py [file.hip]"""
import os, re, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import isa_check  # noqa: E402

VALU = re.compile(r"\s*v_(?!mfma|accvgpr)")


def regs(text):
    out = []
    for a, b in re.findall(r"v\[(\d+):(\d+)\]", text):
        out += list(range(int(a), int(b) + 1))
    out += [int(a) for a in re.findall(r"(?<![\[:\w])v(\d+)\b", text)]
    return out


def scan(text, name_filter="conv_w2d_kernel", result_ws=11):
    """result_ws: wait states behind an MFMA before its result may be read (8-pass: 11; 16-pass: 19)"""
    res = {}
    for blk in re.split(r"\n(?=_Z\w+:)", text):
        name = blk.split(":", 1)[0]
        if name_filter not in name or "s_endpgm" not in blk:
            continue
        lines = blk[:blk.index("s_endpgm")].split("\n")
        mfma_w, valu_w = {}, {}          # register -> wait states since it was written by an MFMA / a VALU instruction
        v1, v2, ex = 0, 0, []
        for l in lines:
            if not re.match(r"\s+[a-z]", l) or l.strip().startswith(";"):
                continue
            ws = 1
            m = re.match(r"\s*s_nop (\d+)", l)
            if m:
                ws = int(m.group(1)) + 1
            ops = l.split(None, 1)[1].split(";")[0] if len(l.split(None, 1)) > 1 else ""
            parts = [x.strip() for x in ops.split(",")]
            mm = re.match(r"\s*v_mfma_", l)
            if mm:
                for r in regs(parts[1]) + regs(parts[2]):          # A, B
                    if valu_w.get(r, 99) < 2:
                        v2 += 1
                        ex.append(("valu->mfma", l.strip(), r, valu_w[r]))
                for r in regs(parts[0]):
                    mfma_w[r] = 0
                    valu_w.pop(r, None)
            else:
                is_store = re.match(r"\s*(global_store|scratch_store|ds_write|buffer_store)", l)
                srcs = parts if is_store else parts[1:]
                for r in regs(",".join(srcs)):
                    if mfma_w.get(r, 99) < result_ws:
                        v1 += 1
                        ex.append(("mfma->read", l.strip(), r, mfma_w[r]))
                if VALU.match(l) and parts:
                    for r in regs(parts[0]):
                        valu_w[r] = 0
                        mfma_w.pop(r, None)
                elif parts and re.match(r"\s*(ds_read|global_load|buffer_load|scratch_load)", l):
                    for r in regs(parts[0]):
                        valu_w.pop(r, None); mfma_w.pop(r, None)
            for d in (mfma_w, valu_w):
                for r in list(d):
                    d[r] += ws
                    if d[r] > 24:
                        del d[r]
        res[name] = (v1, v2, ex[:4])
    return res


if __name__ == "__main__":
    if len(sys.argv) > 1:
        jobs = [(sys.argv[1], "_kernel", 19)]
    else:   # conv_w2d (asm MFMAs) and the fp16-operand kernels (asm v_cvt_pk_f16_f32 in front of builtin 16-pass MFMAs)
        jobs = [(os.path.join(isa_check.CSRC, "conv_w2d_1.hip"), "conv_w2d_kernel", 11), (os.path.join(isa_check.CSRC, "conv_g1_2.hip"), "conv_g1_kernel", 0),
                (os.path.join(isa_check.CSRC, "conv_g1w_4.hip"), "conv_g1w_kernel", 0)]   # (builtin MFMAs: hipcc guards their results itself; 0 = not scanned)
    for src, flt, ws in jobs:
        for k, (v1, v2, ex) in scan(isa_check.compile_asm(src), flt, ws).items():
            print(k[:70], "| MFMA result read early:", v1, "| VALU result read early by an MFMA:", v2, ex)
