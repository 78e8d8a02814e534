"""Builds libfcflow.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree.

    python -m flowcompare_amd.build            # incremental
    python -m flowcompare_amd.build --force
    python -m flowcompare_amd.build --dev      # with the developer kernel variants (-DFC_DEV_VARIANTS; default builds leave them out)

hipcc cross-compiles without a GPU, so this also runs in the CPU-only build container.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libfcflow.so")
ARCH = "gfx950"
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function", "-ffp-contract=off"]
# per-file additions.  premlp.hip: hipcc packs adjacent scalar f32 adds / muls of the epilogue into v_pk_add_f32 / v_pk_mul_f32 (the guide lists
# packed f32 VALU beside MFMAs as an anti-lever; measured here: no difference).  attention.hip: the same packing in the softmax: -1.9 % kernel time
# without it (28.2 -> 27.7 ms per C2 step, same box); gemm.hip is 1 % FASTER with the packing and keeps it.  Developer option for same-box A/B runs:
# FC_EXTRA_FLAGS="attention.hip:-fno-slp-vectorize;gemm.hip:-fno-slp-vectorize" adds flags to single files.
EXTRA_FLAGS = {"premlp.hip": ["-fno-slp-vectorize"], "attention.hip": ["-fno-slp-vectorize"],
               # mlprows.hip: the epilogue is hand-placed in micro-steps behind single MFMAs; SLP packing merges steps of different slots
               "mlprows.hip": ["-fno-slp-vectorize"]}
# (spline_wide.hip: measured both ways on one box, profiles/r04D_*: 68.1 against 68.5 ms per C2 step, training step 1052 against 1050 ms -- no flag)
for _kv in os.environ.get("FC_EXTRA_FLAGS", "").split(";"):
    if ":" in _kv:
        EXTRA_FLAGS.setdefault(_kv.split(":", 1)[0], []).extend(_kv.split(":", 1)[1].split(","))


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp")))


def _newer(a, deps):
    if not os.path.exists(a):
        return False
    t = os.path.getmtime(a)
    return all(os.path.getmtime(d) <= t for d in deps)


def build(force=False, verbose=False, dev=None):
    """dev=True adds -DFC_DEV_VARIANTS: the kernel variants that lost an A/B (csrc/common.h FC_DEV) are compiled in and reachable through
    fc_debug_set; the default library holds the shipped kernels, the bf16-limb range fallback and one fp32-input reference loop.  The choice is
    remembered in _obj/.dev so that an incremental build does not mix objects; switching forces a full rebuild."""
    os.makedirs(OBJ, exist_ok=True)
    marker = os.path.join(OBJ, ".dev")
    was_dev = os.path.exists(marker)
    if dev is None:
        dev = bool(os.environ.get("FC_DEV_VARIANTS"))
    if dev != was_dev:
        force = True
        if dev:
            open(marker, "w").close()
        else:
            os.remove(marker)
    global FLAGS
    FLAGS = [f for f in FLAGS if f != "-DFC_DEV_VARIANTS"] + (["-DFC_DEV_VARIANTS"] if dev else [])
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(HERE, "..", "include", "fcflow.h"))
    jobs = []
    objs = []
    for src in _sources():
        sp = os.path.join(CSRC, src)
        op = os.path.join(OBJ, src + ".o")
        objs.append(op)
        if force or not _newer(op, [sp] + headers):
            cmd = [HIPCC] + FLAGS + EXTRA_FLAGS.get(src, []) + (["-x", "hip"] if src.endswith(".hip") else []) + ["-c", sp, "-o", op]
            jobs.append((src, cmd))

    def run(job):
        src, cmd = job
        r = subprocess.run(cmd, capture_output=True, text=True)
        return src, r.returncode, r.stdout + r.stderr

    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        for src, rc, out in ex.map(run, jobs):
            if verbose or rc != 0:
                sys.stderr.write(f"[hipcc] {src}\n{out}\n")
            if rc != 0:
                raise RuntimeError(f"hipcc failed on {src}")
    if jobs or not os.path.exists(LIB):
        cmd = [HIPCC, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            sys.stderr.write(r.stdout + r.stderr)
            raise RuntimeError("link of libfcflow.so failed")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="-v" in sys.argv, dev=True if "--dev" in sys.argv else None))
