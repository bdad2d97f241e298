"""Guard for the build rule "no packed-FP32 VALU instructions in any kernel" (vcm_ts_amd/csrc/Makefile).

Extracts every gfx950 code object of a shared library (llvm-objdump --offloading), disassembles it and counts
v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32.  Exit status 1 (and a listing per code object) if any is present, or if the
library holds no gfx950 code object at all.  Used by `make check-isa` (part of `make all`) and tests/test_build.py.
usage: check_isa.py path/to/lib.so [more.so ...]"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

OBJDUMP = os.environ.get("LLVM_OBJDUMP", "/opt/rocm/lib/llvm/bin/llvm-objdump")
PACKED = re.compile(r"\bv_pk_(fma|mul|add)_f32\b")
KERNEL = re.compile(r"^[0-9a-f]+ <([^>]+)>:")


def packed_fp32_census(lib_path):
    """{code object: {"instructions": n, "mfma": n, "packed": {kernel symbol: count}}} for every gfx950 object."""
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        local = os.path.join(tmp, os.path.basename(lib_path))
        shutil.copy(lib_path, local)
        subprocess.run([OBJDUMP, "--offloading", local], cwd=tmp, check=True, stdout=subprocess.DEVNULL)
        for name in sorted(os.listdir(tmp)):
            if "gfx950" not in name:
                continue
            dis = subprocess.run([OBJDUMP, "-d", "--mcpu=gfx950", os.path.join(tmp, name)], check=True,
                                 stdout=subprocess.PIPE, text=True).stdout
            sym, packed, n, mfma = "?", {}, 0, 0
            for line in dis.splitlines():
                m = KERNEL.match(line)
                if m:
                    sym = m.group(1)
                    continue
                if "\t" not in line:
                    continue
                n += 1
                mfma += "v_mfma" in line
                if PACKED.search(line):
                    packed[sym] = packed.get(sym, 0) + 1
            out[name] = {"instructions": n, "mfma": mfma, "packed": packed}
    return out


READELF = os.environ.get("LLVM_READELF", "/opt/rocm/lib/llvm/bin/llvm-readelf")


def kernel_resources(lib_path):
    """{kernel symbol: {"vgpr_count", "vgpr_spill_count", "sgpr_count", "group_segment_fixed_size", ...}} from the
    amdhsa metadata notes of every gfx950 code object of the library (what the loader will give each kernel)."""
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        local = os.path.join(tmp, os.path.basename(lib_path))
        shutil.copy(lib_path, local)
        subprocess.run([OBJDUMP, "--offloading", local], cwd=tmp, check=True, stdout=subprocess.DEVNULL)
        for name in sorted(os.listdir(tmp)):
            if "gfx950" not in name:
                continue
            notes = subprocess.run([READELF, "--notes", os.path.join(tmp, name)], check=True, stdout=subprocess.PIPE, text=True).stdout
            # amdhsa.kernels is a YAML list of records with alphabetically sorted keys: a kernel's record starts at
            # .agpr_count and ends at .wavefront_size; .symbol names it (argument records in between have other keys)
            cur = None
            for line in notes.splitlines():
                m = re.match(r"\s*-?\s*\.(\w+):\s*(.*)$", line)
                if not m:
                    continue
                key, val = m.group(1), m.group(2).strip().strip("'")
                if key == "agpr_count":
                    cur = {}
                if cur is None:
                    continue
                if key == "symbol":
                    out[val[:-3] if val.endswith(".kd") else val] = cur
                elif key in ("agpr_count", "vgpr_count", "vgpr_spill_count", "sgpr_count", "sgpr_spill_count",
                             "group_segment_fixed_size", "private_segment_fixed_size", "max_flat_workgroup_size"):
                    cur[key] = int(val)
    return out


def main(argv):
    bad = False
    for path in argv:
        census = packed_fp32_census(path)
        if not census:
            print(f"{path}: no gfx950 code object found", file=sys.stderr)
            bad = True
        for obj, c in census.items():
            total = sum(c["packed"].values())
            print(f"{obj}: {c['instructions']} instructions, {c['mfma']} v_mfma, {total} packed-FP32")
            for sym, k in sorted(c["packed"].items()):
                print(f"    {k:5d}  {sym}", file=sys.stderr)
            bad = bad or total > 0
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
