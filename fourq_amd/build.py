"""Builds libfourq_amd.so (HIP, gfx950 only) in-tree with hipcc.

    python -m fourq_amd.build            # or: from fourq_amd.build import build_library

hipcc cross-compiles without a GPU, so this runs in the build container; the resulting .so is
git-ignored but travels to the GPU box with the tree.

Every compile also asks hipcc for its per-kernel resource report (-Rpass-analysis=kernel-resource-usage); the
report of the product build is written to fourq_amd/kernel_resources.json and checked against RESOURCE_POLICY below,
so a compiler or source change that makes a hot kernel spill (or lose its occupancy) fails the build here, on the CPU,
instead of showing up as a slower -- or, for a kernel with hand-placed waits, wrong -- GPU run.
"""
import json
import os
import re
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC_DIR = os.path.join(HERE, "csrc")
LIB_PATH = os.path.join(HERE, "libfourq_amd.so")
RESOURCES_PATH = os.path.join(HERE, "kernel_resources.json")
PLACEMENT_PATH = os.path.join(HERE, "code_placement.json")
# four translation units: FQ_CHAIN=0 / 1 (kernels.hip.h), and the constant-time-selection builds of both flavours
SOURCES = ["fourq_amd.hip", "fourq_chain.hip", "fourq_ct_fused.hip", "fourq_ct_chain.hip"]
HEADERS = ["fp127.hip.h", "curve.hip.h", "recode.hip.h", "kernels.hip.h", "pair.hip.h", "ladder_asm.hip.h", "ladder_asm_gfx950.inc", "constants.inc", "pipeline_plan.h", os.path.join("..", "..", "include", "fourq_amd.h")]
HIPCC_FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Rpass-analysis=kernel-resource-usage"]
# Code placement (tools/asmgen/place_asm.py, profiles/r04_ladder_step.txt): the device code of every translation unit goes through
# assembly text, where every 8-byte instruction is put on an 8-byte boundary (an _e32 instruction in front of it re-encoded as _e64),
# and is assembled, linked and bundled by the same tools hipcc drives itself (`hipcc -###` prints the four steps).
PLACE_TOOL = os.path.normpath(os.path.join(HERE, "..", "tools", "asmgen", "place_asm.py"))
LLVM_BIN = "/opt/rocm/lib/llvm/bin"
BUNDLE_TARGETS = "host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950"
BUNDLE_COMPRESS = [] if os.environ.get("FOURQ_BUILD_NO_COMPRESS") == "1" else ["--compress"]

# (regex on the demangled kernel name, field, predicate, why).  Checked for the product build only (no extra flags).
RESOURCE_POLICY = [
    (r"ladder_kernel<\d, 2,", "scratch", lambda v: v == 0, "the two-kernel route's ladders must not spill"),
    (r"ladder_kernel<\d, 2, true", "occupancy", lambda v: v >= 4, "the two-kernel route's DH ladders are sized for 4 waves per SIMD (128 VGPRs)"),
    (r"ladder_kernel<\d, 2, false", "occupancy", lambda v: v >= 2, "the two-kernel route's MUL ladders run the asm bodies at 2 waves per SIMD"),
    (r"ladder_kernel<\d, 1, \w+, \w+, false>", "scratch", lambda v: v == 0, "the fixed-base ladders (MUL and DH) must not spill"),
    (r"ladder_kernel<\d, 1, \w+, \w+, false>", "occupancy", lambda v: v >= 2, "the fixed-base ladders run the asm bodies at 2 waves per SIMD"),
    (r"ladder_kernel<\d, 1, \w+, \w+, true>", "scratch", lambda v: v == 0, "the constant-time fixed-base ladders must not spill"),
    (r"ladder_kernel<\d, 0,", "scratch", lambda v: v == 0, "the fused variable-base kernels must not spill to memory (AGPR copies are fine)"),
    (r"pair_kernel<", "scratch", lambda v: v == 0, "the two-lanes-per-element kernel must not spill"),
    (r"prep_kernel<0, \w+>\(", "occupancy", lambda v: v >= 2, "prep_kernel<ENDO> hides its read-backs behind a second wave per SIMD"),
    (r"mixed_ct_tail_kernel<", "occupancy", lambda v: v >= 2, "the constant-time mixed-batch tail runs the asm bodies at two waves per SIMD"),
    (r"mixed_ct_tail_kernel<", "scratch", lambda v: v == 0, "the constant-time mixed-batch tail must not spill"),
    (r"comb_kernel<true, false>", "scratch", lambda v: v == 0, "the keygen comb of large batches (deferred normalisation) must not touch scratch memory"),
    (r"comb_kernel<false, false>", "scratch", lambda v: v <= 16, "the keygen comb with its inversion in the kernel parks 4 registers around the call: no more than that"),
]


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the FourQ engine needs the ROCm toolchain to build")
    return exe


def _demangle(names):
    filt = shutil.which("c++filt") or "/opt/rocm/lib/llvm/bin/llvm-cxxfilt"
    try:
        out = subprocess.run([filt], input="\n".join(names), capture_output=True, text=True, check=True).stdout.splitlines()
        return [o.replace("fq::(anonymous namespace)::", "").replace("(anonymous namespace)::", "") for o in out]
    except (OSError, subprocess.CalledProcessError):
        return list(names)


_FIELDS = {"TotalSGPRs": "sgprs", "VGPRs": "vgprs", "AGPRs": "agprs", "ScratchSize [bytes/lane]": "scratch",
           "Occupancy [waves/SIMD]": "occupancy", "VGPRs Spill": "vgpr_spills", "SGPRs Spill": "sgpr_spills", "LDS Size [bytes/block]": "lds"}


def parse_resource_remarks(text):
    """hipcc -Rpass-analysis=kernel-resource-usage output -> {mangled kernel name: {field: int}}."""
    kernels, cur = {}, None
    for line in text.splitlines():
        m = re.search(r"remark:\s+(Function Name|[A-Za-z ]+(?:\[[^\]]+\])?):\s*(\S+)\s*\[-Rpass-analysis", line)
        if not m:
            continue
        key, val = m.group(1).strip(), m.group(2)
        if key == "Function Name":
            cur = kernels.setdefault(val, {})
        elif cur is not None and key in _FIELDS:
            try:
                cur[_FIELDS[key]] = int(val)
            except ValueError:
                pass
    return kernels


def check_policy(resources):
    """Violations of RESOURCE_POLICY in {unit: {demangled kernel: fields}} as a list of strings."""
    bad = []
    for unit, kernels in resources.items():
        for name, f in kernels.items():
            for pattern, field, ok, why in RESOURCE_POLICY:
                if re.search(pattern, name) and field in f and not ok(f[field]):
                    bad.append("%s: %s has %s = %d (%s)" % (unit, name, field, f[field], why))
    return bad


def source_id(extra_flags=()):
    """16 hex digits identifying what a build was made from: the translation units, their headers and the flags.  Compiled into
    the library (fourq_build_id) so that a profile can say which build it was taken on (bench.py: roofline.traffic_source)."""
    import hashlib
    h = hashlib.sha256()
    for f in [os.path.join(SRC_DIR, f) for f in SOURCES + HEADERS] + [PLACE_TOOL]:
        with open(os.path.normpath(f), "rb") as fh:
            h.update(fh.read())
    h.update(" ".join(list(HIPCC_FLAGS) + list(extra_flags)).encode())
    return h.hexdigest()[:16]


def is_stale():
    if not os.path.exists(LIB_PATH):
        return True
    built = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(SRC_DIR, f) for f in SOURCES + HEADERS] + [PLACE_TOOL]
    return any(os.path.getmtime(os.path.normpath(d)) > built for d in deps)


FASTCODEC_SRC = os.path.join(SRC_DIR, "fastcodec.c")


def fastcodec_path():
    import sysconfig
    return os.path.join(HERE, "_fastcodec" + sysconfig.get_config_var("EXT_SUFFIX"))


def build_fastcodec(force=False, verbose=False):
    """The host-side tuple <-> word packer (csrc/fastcodec.c, a CPython extension; no GPU code) built in-tree with gcc."""
    import sysconfig
    out = fastcodec_path()
    if not force and os.path.exists(out) and os.path.getmtime(out) >= os.path.getmtime(FASTCODEC_SRC):
        return out
    cc = shutil.which("gcc") or shutil.which("cc")
    if cc is None:
        raise RuntimeError("no C compiler for fourq_amd/csrc/fastcodec.c")
    _run([cc, "-O2", "-shared", "-fPIC", "-Wall", "-I" + sysconfig.get_paths()["include"], "-o", out, FASTCODEC_SRC], verbose)
    return out


def _run(cmd, verbose=False):
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if proc.returncode != 0:
        raise RuntimeError("command failed: %s\n%s" % (" ".join(cmd), "\n".join(l for l in proc.stdout.splitlines() if "remark:" not in l)[-6000:]))
    return proc.stdout


class RegisterOwnershipError(Exception):
    """A kernel reaches a hand-scheduled body whose physical registers lie outside its allocation.  NOT a RuntimeError on purpose: the
    build's fallback for a failed placement detour catches RuntimeError, and this finding must never be swallowed by it -- the plain
    hipcc build of the same sources would carry the same kernel."""


def check_register_ranges(asm_path):
    """The generated bodies (ladder_asm.hip.h) name their temporaries as PHYSICAL registers (v168-v255, v236-v255) and declare them as
    clobbers.  hipcc counts clobbered registers into a kernel's allocation -- unless the kernel's launch bounds cap its budget below them, in
    which case it only warns ("reserved registers") and the instruction would address registers the wave does not own.  No such kernel may
    reach a body: every kernel's highest VGPR named in its code must lie inside its ARCHITECTURAL VGPR allocation.  On gfx950
    `.amdhsa_next_free_vgpr` is the unified VGPR + AGPR total; where a kernel has AGPRs, `.amdhsa_accum_offset` is where they begin, i.e.
    the number of arch VGPRs (a v236 named by a kernel with accum_offset 176 would alias its AGPRs and still be below next_free_vgpr).
    A kernel that names VGPRs but has no parsable allocation fails too."""
    kernel, highest, alloc, accum = None, {}, {}, {}
    with open(asm_path) as fh:
        for ln in fh:
            m = re.match(r"^(_Z\w+):", ln)
            if m:
                kernel = m.group(1)
                continue
            m = re.match(r"\s*\.amdhsa_kernel\s+(\S+)", ln)
            if m:
                kernel = m.group(1)
                continue
            m = re.match(r"\s*\.amdhsa_next_free_vgpr\s+(\d+)", ln)
            if m and kernel:
                alloc[kernel] = int(m.group(1))
                continue
            m = re.match(r"\s*\.amdhsa_accum_offset\s+(\d+)", ln)
            if m and kernel:
                accum[kernel] = int(m.group(1))
                continue
            t = ln.strip()
            if kernel is None or not t or t[0] in ";.":
                continue
            for a, b in re.findall(r"\bv(\d+)\b|\bv\[\d+:(\d+)\]", t):
                r = int(a or b)
                if r > highest.get(kernel, -1):
                    highest[kernel] = r
    bad = []
    for k, h in highest.items():
        if k not in alloc:
            # device functions that are not kernels have no descriptor; only entry points (.amdhsa_kernel) are checked
            continue
        owned = min(alloc[k], accum.get(k, alloc[k]))      # accum_offset < next_free_vgpr  <=>  the kernel has AGPRs above its arch VGPRs
        if h >= owned:
            bad.append("%s names v%d but owns %d arch VGPRs (next_free_vgpr %d, accum_offset %s)" % (k, h, owned, alloc[k], accum.get(k, "-")))
    if not alloc and highest:
        bad.append("%s: VGPRs are named but no .amdhsa_next_free_vgpr could be parsed -- the listing format changed" % asm_path)
    if bad:
        raise RegisterOwnershipError("a kernel reaches an asm body whose registers it does not own:\n  " + "\n  ".join(bad))


def compile_unit(src, obj, flags, verbose=False, place=True, placement=None):
    """One translation unit -> object file; returns hipcc's remarks (the kernel resource report).  With `place`, the device code
    takes the detour through placed assembly text described at PLACE_TOOL; `placement[src]` receives what the pass did."""
    src_path = os.path.join(SRC_DIR, src)
    stem = os.path.splitext(obj)[0]
    if not place:
        if placement is not None:
            placement[src] = {"placed": False}
        # the register-ownership check reads the device listing, so the unplaced build makes one too (the fallback and the
        # -DFQ_NO_PLACE=1 variants used to skip the check); the object itself comes from the ordinary one-step compile
        listing = stem + ".check.s"
        try:
            _run([_hipcc()] + [f for f in flags if not f.startswith("-Rpass")] + ["--cuda-device-only", "-S", "-o", listing, src_path], verbose)
            check_register_ranges(listing)
        finally:
            if os.path.exists(listing):
                os.remove(listing)
        return _run([_hipcc()] + flags + ["-c", "-o", obj, src_path], verbose)
    import importlib.util
    spec = importlib.util.spec_from_file_location("place_asm", PLACE_TOOL)
    place_asm = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(place_asm)
    dev_s, placed_s, dev_o, hsaco, fatbin = stem + ".dev.s", stem + ".placed.s", stem + ".dev.o", stem + ".hsaco", stem + ".hipfb"
    remarks = _run([_hipcc()] + flags + ["--cuda-device-only", "-S", "-o", dev_s, src_path], verbose)
    check_register_ranges(dev_s)
    stats = place_asm.place_file(dev_s, placed_s)
    if placement is not None:
        placement[src] = {"placed": True, "wide_instructions": stats["wide_total"], "at_4_mod_8_before": stats["misaligned_before"],
                          "at_4_mod_8_after": stats["misaligned_after"], "reencoded": stats["promoted"], "nops": stats["nops"],
                          "instructions_compared_with_the_unplaced_object": stats["instructions_compared"]}
    if verbose:
        print("%s: %d of %d 8-byte instructions at 4 mod 8 before placement, %d after" % (
            src, stats["misaligned_before"], stats["wide_total"], stats["misaligned_after"]), file=sys.stderr)
    _run([os.path.join(LLVM_BIN, "clang"), "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", placed_s, "-o", dev_o], verbose)
    _run([os.path.join(LLVM_BIN, "lld"), "-flavor", "gnu", "-m", "elf64_amdgpu", "--no-undefined", "-shared", "-o", hsaco, dev_o], verbose)
    # --compress: the four code objects are 5 MB of the 5.2 MB library, nearly all of it generated straight-line code; the HIP runtime
    # unpacks a compressed bundle when it loads the module (once per process, a few milliseconds).  1.4 MB instead of 5.2 to push to a GPU box.
    _run([os.path.join(LLVM_BIN, "clang-offload-bundler"), "-type=o", "-bundle-align=4096", "-targets=" + BUNDLE_TARGETS] + BUNDLE_COMPRESS +
         ["-input=/dev/null", "-input=" + hsaco, "-output=" + fatbin], verbose)
    host_flags = [f for f in flags if not f.startswith("-Rpass")]
    _run([_hipcc()] + host_flags + ["--cuda-host-only", "-Xclang", "-fcuda-include-gpubinary", "-Xclang", fatbin, "-c", "-o", obj, src_path], verbose)
    for tmp in (dev_s, placed_s, dev_o, hsaco, fatbin):
        os.remove(tmp)
    return remarks


def build_library(force=False, verbose=False, extra_flags=(), out_path=None):
    """Compile the library if it is missing or older than its sources; returns its path.  `out_path` + `extra_flags`
    build an experiment variant beside the product library (tools/ab_bench.sh compares them on one GPU box); a variant's
    resource report goes to <out_path>.resources.json and is not policy-checked.  The pseudo-flag -DFQ_NO_PLACE=1 builds a
    variant without the code placement pass."""
    if out_path is None:
        try:
            build_fastcodec(force, verbose)
        except Exception as exc:            # optional: codec.py has a pure-Python path.  A host without gcc, or a CPython whose int
            #                                 internals fastcodec.c does not know, must not stop the GPU library from being built
            print("fourq_amd.build: WARNING -- the tuple codec extension (csrc/fastcodec.c) did not build; fourq_amd.codec falls back to "
                  "its pure-Python conversions (~5 us per element instead of ~0.5):\n%s" % str(exc)[-1500:], file=sys.stderr)
    if out_path is None and not force and not is_stale():
        return LIB_PATH
    from concurrent.futures import ThreadPoolExecutor
    lib_path = out_path or LIB_PATH
    suffix = "" if out_path is None else "." + os.path.splitext(os.path.basename(out_path))[0]
    place = "-DFQ_NO_PLACE=1" not in extra_flags
    flags = HIPCC_FLAGS + list(extra_flags) + ['-DFQ_BUILD_ID="%s"' % source_id(extra_flags)]
    objs = [os.path.join(SRC_DIR, os.path.splitext(src)[0] + suffix + ".o") for src in SOURCES]
    placement = {}
    def one(so):
        try:
            return compile_unit(so[0], so[1], flags, verbose, place, placement)
        except RuntimeError as e:
            if not place:
                raise
            # the placement detour failed (an assembler that rejects a re-encoding, a tool that moved): the plain hipcc build of the
            # same sources is correct, only a percent slower -- build that, and say so loudly.  A RegisterOwnershipError is not a
            # RuntimeError and passes straight through: it is a finding about the kernel, not about the detour
            print("fourq_amd.build: code placement of %s failed, building it without the pass:\n%s" % (so[0], str(e)[-1500:]), file=sys.stderr)
            return compile_unit(so[0], so[1], flags, verbose, False, placement)

    with ThreadPoolExecutor(max_workers=len(SOURCES)) as pool:      # the translation units in parallel (~1 min each)
        outs = list(pool.map(one, zip(SOURCES, objs)))
    resources = {}
    for src, out in zip(SOURCES, outs):
        raw = parse_resource_remarks(out)
        names = list(raw)
        resources[src] = dict(zip(_demangle(names), (raw[n] for n in names)))
    bad = check_policy(resources) if out_path is None and not extra_flags else []
    if bad:
        raise RuntimeError("kernel resource policy violated (fourq_amd/build.py RESOURCE_POLICY):\n  " + "\n  ".join(bad))
    link = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib_path] + objs
    if verbose:
        print(" ".join(link), file=sys.stderr)
    proc = subprocess.run(link, capture_output=True, text=True)
    if proc.returncode != 0:
        raise RuntimeError("link failed:\n" + proc.stdout + proc.stderr)
    with open(RESOURCES_PATH if out_path is None else lib_path + ".resources.json", "w") as fh:
        json.dump(resources, fh, indent=1, sort_keys=True)
    with open(PLACEMENT_PATH if out_path is None else lib_path + ".placement.json", "w") as fh:      # what the placement pass did, per unit
        json.dump(placement, fh, indent=1, sort_keys=True)
    if out_path is not None:
        for obj in objs:                                  # a variant's objects have served their purpose
            os.remove(obj)
    return lib_path


if __name__ == "__main__":
    # python -m fourq_amd.build [--force] [--out variants/libX.so -DFLAG=1 ...]
    argv = sys.argv[1:]
    out = argv[argv.index("--out") + 1] if "--out" in argv else None
    flags = [a for a in argv if a.startswith("-D") or a.startswith("-m")]
    print(build_library(force="--force" in argv, verbose=True, extra_flags=flags, out_path=os.path.abspath(out) if out else None))
