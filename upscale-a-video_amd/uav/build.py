"""Build libuav_hip.so (gfx950) in-tree with hipcc.

The library is built next to this file so that it travels with the source snapshot to the GPU
box (the .so is git-ignored, not gpurun-ignored).  hipcc cross-compiles without a GPU.
"""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(HERE), "csrc")
LIB = os.path.join(HERE, "libuav_hip.so")
STAMP = os.path.join(HERE, ".libuav_hip.stamp")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value"]


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _digest():
    h = hashlib.sha256()
    files = sources() + [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".h")]
    files.append(os.path.join(os.path.dirname(os.path.dirname(HERE)), "include", "uav_hip.h"))
    for f in files:
        with open(f, "rb") as fh:
            h.update(os.path.basename(f).encode()); h.update(fh.read())     # path-independent: the tree is copied to other roots
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def conv_kernel_digest():
    """Digest of the sources of the dominant kernel family only (csrc/conv_*.hip / conv_*.h of the fp16 implicit GEMM + the shared device header): what a PMC traffic
    pass of the conv kernels stays valid for (bench.py replays profiles/pmc_conv_traffic.json only on a match)."""
    h = hashlib.sha256()
    conv = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.startswith("conv_") and f.endswith((".hip", ".h")) and "f32" not in f)
    for f in conv + [os.path.join(CSRC, "uav_common.h")]:
        with open(f, "rb") as fh:
            h.update(os.path.basename(f).encode()); h.update(fh.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def is_fresh():
    if not (os.path.exists(LIB) and os.path.exists(STAMP)):
        return False
    with open(STAMP) as fh:
        return fh.read().strip() == _digest()


def build(force=False, verbose=True):
    """Compile every csrc/*.hip for gfx950 and link libuav_hip.so.  Returns the library path."""
    if not force and is_fresh():
        return LIB
    if not os.path.exists(HIPCC):
        raise RuntimeError(f"hipcc not found at {HIPCC}; cannot build libuav_hip.so")
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    # one builder at a time: with one process per GPU all ranks import the package at once
    import fcntl
    lock = open(os.path.join(objdir, ".lock"), "w")
    fcntl.flock(lock, fcntl.LOCK_EX)
    try:
        if not force and is_fresh():     # another rank built it while we waited
            return LIB
        return _build_locked(objdir, verbose)
    finally:
        fcntl.flock(lock, fcntl.LOCK_UN)
        lock.close()


def _build_locked(objdir, verbose):

    def cc(src):
        obj = os.path.join(objdir, os.path.basename(src)[:-4] + ".o")
        cmd = [HIPCC] + FLAGS + ["-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
        return obj

    with ThreadPoolExecutor(max_workers=8) as ex:
        asm_jobs = {f: ex.submit(_device_assembly, f) for f in NAMED_ACC_KERNELS}      # the audit's `hipcc -S` runs beside the compiles
        objs = list(ex.map(cc, sources()))
        asm = {f: j.result() for f, j in asm_jobs.items()}
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    audit_accumulator_file(asm)
    audit_conv_scratch()
    audit_no_scratch()
    with open(STAMP, "w") as fh:
        fh.write(_digest())
    if verbose:
        print(f"[uav] built {LIB} from {len(objs)} objects", file=sys.stderr)
    return LIB


NAMED_ACC_KERNELS = {"attention.hip": ["attn512w_kernel"],
                     "attn512x.hip": ["attn512x_kernel"],
                     "xattn_fused.hip": ["xattn_sublayer_kernelILi0E", "ff_sublayer_kernel"],
                     "tattn_fused.hip": ["tattn_sublayer_kernelILi0ELi0ELi0E", "tattn_sublayer_kernelILi2ELi0ELi0E"],
                     "tattn_block_fused.hip": ["tattn_sublayer_kernelILi2ELi1ELi0E"],
                     "tattn_block_pi_fused.hip": ["tattn_sublayer_kernelILi2ELi1ELi1E"]}


def audit_accumulator_file(asm=None):
    """attn512w_kernel (csrc/attention.hip) and the fused sub-layer kernels (csrc/xattn_fused.hip, tattn_fused.hip, tattn_block_fused.hip) keep their
    256 fp32 accumulators in a[0:255] BY NAME from inline asm.  That is only sound while hipcc itself never touches the accumulator file in
    those kernels (it would treat the registers as free between our statements): compile the file to assembly (once per file: `asm` =
    {file: text} when the caller already did) and fail the build if any compiler-generated instruction of the kernel names an AGPR."""
    if asm is None:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(len(NAMED_ACC_KERNELS)) as ex:
            asm = dict(zip(NAMED_ACC_KERNELS, ex.map(_device_assembly, NAMED_ACC_KERNELS)))
    for fname, kernels in NAMED_ACC_KERNELS.items():
        for k in kernels:
            _audit_named_accumulators(fname, k, asm[fname])


def _device_assembly(fname):
    src = os.path.join(CSRC, fname)
    r = subprocess.run([HIPCC] + FLAGS + ["-S", "--cuda-device-only", "-o", "-", src], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc -S failed on {src}:\n{r.stderr}")
    return r.stdout


def _audit_named_accumulators(fname, kernel, text):
    lines = text.splitlines()
    inside = in_asm = False
    bad = []
    asm_blocks = 0
    for ln in lines:
        if kernel in ln and not ln[:1].isspace() and ln.split(";")[0].rstrip().endswith(":"):
            inside = True
            continue
        if not inside:
            continue
        # the function ends at its .Lfunc_end label / .size directive, not at the first s_endpgm (early-exit branches
        # emit several)
        if ln.startswith(".Lfunc_end") or (ln.lstrip().startswith(".size") and kernel in ln):
            break
        if "#ASMSTART" in ln:
            in_asm = True
            asm_blocks += 1
        elif "#ASMEND" in ln:
            in_asm = False
        elif not in_asm and not ln.lstrip().startswith(";") and ("v_accvgpr" in ln or " a[" in ln or ",a[" in ln):
            bad.append(ln.strip())
    if not inside:
        raise RuntimeError(f"audit: {kernel} not found in the assembly of {fname}")
    if asm_blocks == 0:
        raise RuntimeError(f"audit: no inline-asm block seen inside {kernel} — the scan did not cover the kernel body")
    if bad:
        raise RuntimeError(f"audit: hipcc uses the accumulator file inside {kernel}, which names a[0:255] from inline asm:\n  "
                           + "\n  ".join(bad[:8]))


def _readelf():
    """llvm-readelf: $UAV_READELF, else <rocm>/lib/llvm/bin next to hipcc, else whatever PATH offers."""
    import shutil
    cands = [os.environ.get("UAV_READELF"), os.path.join(os.path.dirname(os.path.dirname(os.path.realpath(HIPCC))), "lib", "llvm", "bin", "llvm-readelf"),
             os.path.join(os.path.dirname(os.path.dirname(HIPCC)), "lib", "llvm", "bin", "llvm-readelf"),
             shutil.which("llvm-readelf")]
    for c in cands:
        if c and os.path.exists(c):
            return c
    raise RuntimeError("llvm-readelf not found (set UAV_READELF): the build audit reads the code-object metadata with it; tried "
                       + ", ".join(str(c) for c in cands if c))


def kernel_metadata(lib=None):
    """{kernel symbol: {metadata field: value}} of every gfx950 code object bundled in the library, read from the AMDGPU
    metadata note (llvm-readelf --notes) of each offload bundle entry."""
    import re
    import struct
    import tempfile
    readelf = _readelf()
    with open(lib or LIB, "rb") as fh:
        blob = fh.read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    out = {}
    pos = 0
    while True:
        i = blob.find(magic, pos)
        if i < 0:
            break
        pos = i + 1
        (entries,) = struct.unpack_from("<Q", blob, i + len(magic))
        if entries > 16:
            continue
        o = i + len(magic) + 8
        for _ in range(entries):
            off, size, tlen = struct.unpack_from("<QQQ", blob, o)
            o += 24
            triple = blob[o:o + tlen].decode(errors="replace")
            o += tlen
            if "gfx950" not in triple or size == 0:
                continue
            with tempfile.NamedTemporaryFile(suffix=".elf") as tf:
                tf.write(blob[i + off:i + off + size]); tf.flush()
                r = subprocess.run([readelf, "--notes", tf.name], capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError(f"llvm-readelf failed on a bundled code object:\n{r.stderr}")
            cur = None
            for ln in r.stdout.splitlines():
                m = re.match(r"\s+(- )?\.(\w+):\s+(\S+)", ln)
                if not m:
                    continue
                if m.group(1) and ln.startswith("  - "):      # a new entry of amdhsa.kernels (args entries are indented deeper)
                    cur = {}
                if cur is None:
                    continue
                cur[m.group(2)] = m.group(3)
                if m.group(2) == "name":
                    out[m.group(3)] = cur
    return out


def audit_no_scratch():
    """every kernel SHIPPED in libuav_hip.so runs without a private segment and without a spilled VGPR (the legacy
    instances that had scratch — attn512_kernel, attn_kernel<512>, the round-1 / LayerNorm-fold conv instances — are development
    kernels now, -DUAV_DEV_KERNELS)."""
    bad = [f"{n}: scratch {md.get('private_segment_fixed_size')} B, {md.get('vgpr_spill_count')} spilled VGPRs"
           for n, md in kernel_metadata().items()
           if int(md.get("private_segment_fixed_size", "0")) != 0 or int(md.get("vgpr_spill_count", "0")) != 0]
    if bad:
        raise RuntimeError("audit: shipped kernels must not use scratch:\n  " + "\n  ".join(bad))


def audit_conv_scratch():
    """The rotated k-step of conv_gemm256i_kernel<6,*,0> and the one-statement k-step of conv_gemm256w_kernel carry LDS-DMA
    targets, staged constants and fragment registers across inline-asm statements with hand-counted vmcnt/lgkmcnt waits
    (csrc/conv_gemm.hip).  A compiler spill of one of those registers to scratch between the statements would read stale
    data without any diagnostic, so the build fails if either kernel family has a private segment or a spilled VGPR
    (SGPR spills go to VGPR lanes and are harmless)."""
    meta = kernel_metadata()
    seen = 0
    bad = []
    for name, md in meta.items():
        if "conv_gemm256w_kernel" in name or "conv_gemm256i_kernelILi6E" in name:
            seen += 1
            if int(md.get("private_segment_fixed_size", "1")) != 0 or int(md.get("vgpr_spill_count", "1")) != 0:
                bad.append(f"{name}: scratch {md.get('private_segment_fixed_size')} B, {md.get('vgpr_spill_count')} spilled VGPRs")
    if seen < 9:          # conv_gemm256i_kernel<6, 0..3> + conv_gemm256w_kernel<0..3> + its hi | lo instance
        raise RuntimeError(f"audit: only {seen} conv_gemm256i<6,*>/conv_gemm256w instances found in {LIB}")
    if bad:
        raise RuntimeError("audit: the asm-scheduled conv kernels must not spill:\n  " + "\n  ".join(bad))


if __name__ == "__main__":
    build(force="--force" in sys.argv)
