#!/usr/bin/env python3
"""The ONE definition of the extractor kernels' vector-issue roof (VERDICT r05 #4b).

    issue_frac(kernel) = SQ_INSTS_VALU (wave-instructions per launch, counted by the hardware in the run that is priced)
                         x cycles_per_instruction(kernel)
                         / (1024 SIMDs x 2.4 GHz x launch duration)

cycles_per_instruction(kernel) = the mean, over the vector instructions of the kernel's code object as shipped (llvm-objdump of
liborbfe.so's gfx950 code object: the STATIC instruction mix, taken as a proxy of the dynamic one -- the hot loops of these
kernels are straight-line code that dominates both), of the issue cost of each opcode as MEASURED on this chip by
tools/valu_rate.hip (profiles/r01_valu_rate.txt: time per wave64 instruction of a long dependent-free stream on every SIMD,
expressed in cycles at 2.4 GHz -- so the clock cancels: the product is a time).  Opcodes the microbenchmark did not cover take
the cost of their class (the simple 32-/16-bit integer, logic, shift-right, move and f32 mul/fma class ~2.7 cycles; every other
VALU opcode ~4.5; DPP / SDWA forms 4.5; f64 16; v_pk_*_f32 and v_max3_i16 8.5).

usage: tools/isa/issue_table.py [liborbfe.so] > profiles/r06_issue_table.json
bench.py imports issue_table() and prices the dominant kernel with it in every run."""
import collections
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
LLVM = "/opt/rocm/lib/llvm/bin"
FAST_CLASS = {'v_add_u32', 'v_sub_u32', 'v_subrev_u32', 'v_and_b32', 'v_or_b32', 'v_xor_b32', 'v_lshrrev_b32', 'v_mov_b32', 'v_add_u16',
              'v_sub_u16', 'v_min_i16', 'v_max_i16', 'v_min_u16', 'v_max_u16', 'v_bitop3_b32', 'v_mul_f32', 'v_fma_f32', 'v_fmac_f32',
              'v_add_f32', 'v_sub_f32', 'v_not_b32', 'v_ashrrev_i32', 'v_bitop3_b16', 'v_mul_lo_u16', 'v_lshrrev_b16',
              'v_accvgpr_write_b32', 'v_accvgpr_read_b32'}


def measured_costs(path=os.path.join(ROOT, "profiles", "r01_valu_rate.txt")):
    """opcode -> cycles per wave64 instruction at 2.4 GHz, as measured (profiles/r01_valu_rate.txt)."""
    out = {}
    for ln in open(path):
        m = re.match(r"^(v_[a-z0-9_]+)(\([a-z]+\))?\s+[\d.]+ ms\s+([\d.]+) cycles/wave-instr", ln)
        if m:
            out[m.group(1)] = float(m.group(3))
    return out


def opcode_cost(op, measured):
    base = re.sub(r"_(e32|e64)$", "", op)
    # (v_cndmask_b32 reads 22.6 cycles in the microbenchmark: every instruction of its stream reads VCC, which the stream's own
    # v_cmp rewrites -- a dependency stall, not an issue cost; it is priced with its class)
    if base in measured and base != "v_cndmask_b32":
        return measured[base], "measured"
    stem = re.sub(r"_(dpp|sdwa)$", "", base)
    if base.endswith("_dpp") or base.endswith("_sdwa"):
        return measured.get(base, 4.5), "class:dpp/sdwa"
    if stem in FAST_CLASS:
        return 2.7, "class:simple"
    if stem.startswith("v_pk_") and stem.endswith("_f32") or stem == "v_max3_i16":
        return 8.5, "class:8"
    if "f64" in stem:
        return 16.0, "class:f64"
    if stem.startswith("v_mfma"):
        return 8.0, "class:mfma-issue"
    return 4.5, "class:other"


def disassemble(lib):
    tmp = tempfile.mkdtemp(prefix="orbfe_isa_")
    try:
        so = os.path.join(tmp, "lib.so")
        shutil.copy(lib, so)  # (llvm-objdump --offloading writes the code objects next to its input)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", so], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        text = ""
        for f in sorted(os.listdir(tmp)):
            if "gfx950" in f:
                text += subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", os.path.join(tmp, f)], check=True,
                                       stdout=subprocess.PIPE, text=True).stdout
        return text
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), stdout=subprocess.PIPE, text=True, check=True).stdout
        return dict(zip(names, out.split("\n")))
    except Exception:  # noqa: BLE001
        return {n: n for n in names}


def issue_table(lib=None):
    lib = lib or os.path.join(ROOT, "orb_slam3_detailed_comments_kor_amd", "liborbfe.so")
    measured = measured_costs()
    text = disassemble(lib)
    per = collections.OrderedDict()
    cur = None
    for ln in text.split("\n"):
        m = re.match(r"^[0-9a-f]+ <([^>]+)>:", ln)
        if m:
            cur = per.setdefault(m.group(1), collections.Counter())
            continue
        s = ln.strip()
        if cur is None or not s:
            continue
        op = s.split()[0]
        if re.match(r"^(v_|s_|ds_|global_|buffer_|flat_|scratch_)", op):
            cur[op] += 1
    names = demangle(list(per))
    table = {}
    for mangled, cnt in per.items():
        name = names[mangled].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
        if not name.startswith("k_"):
            continue
        valu = {op: n for op, n in cnt.items() if op.startswith("v_") and not op.startswith("v_mfma")}
        mfma = sum(n for op, n in cnt.items() if op.startswith("v_mfma"))
        nv = sum(valu.values())
        if nv == 0:
            continue
        cyc = 0.0
        how = collections.Counter()
        rows = []
        for op, n in sorted(valu.items(), key=lambda kv: -kv[1]):
            c, h = opcode_cost(op, measured)
            cyc += c * n
            how[h.split(":")[0]] += n
            rows.append([op, n, c, h])
        table[name] = {"static_valu_instructions": nv, "static_mfma_instructions": mfma,
                       "static_salu": sum(n for op, n in cnt.items() if op.startswith("s_")),
                       "static_lds": sum(n for op, n in cnt.items() if op.startswith("ds_")),
                       "static_vmem": sum(n for op, n in cnt.items() if re.match(r"^(global_|buffer_|flat_|scratch_)", op)),
                       "cycles_per_instruction": cyc / nv,
                       "share_priced_by_measurement": how["measured"] / nv,
                       "opcodes": rows[:24]}
    return {"definition": "issue_frac = SQ_INSTS_VALU x cycles_per_instruction / (1024 SIMDs x 2.4 GHz x launch duration); "
                          "cycles_per_instruction = static-mix mean of the per-opcode issue costs measured in profiles/r01_valu_rate.txt "
                          "(cycles at 2.4 GHz per wave64 instruction; classes for opcodes it does not list)",
            "source": "llvm-objdump -d of the gfx950 code objects of " + os.path.relpath(lib, ROOT),
            "kernels": table}


if __name__ == "__main__":
    print(json.dumps(issue_table(sys.argv[1] if len(sys.argv) > 1 else None), indent=1))
