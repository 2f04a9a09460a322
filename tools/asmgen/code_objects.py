"""The gfx950 code objects inside a host library's .hip_fatbin section, one ELF file each (build box; used by tests/test_place_asm.py).

The build ships them as COMPRESSED offload bundles (fourq_amd/build.py: `clang-offload-bundler --compress`), which `llvm-objdump --offloading`
does not unpack: the section is cut at the bundles' magic words, a compressed bundle trimmed to the length its header states (the section
pads each to 4 KiB) and unbundled.  (Kept out of place_asm.py on purpose: that file is hashed into the library's build id.)"""
import os
import re
import struct
import subprocess

LLVM = "/opt/rocm/lib/llvm/bin"


def code_objects(lib, out_dir, target="hipv4-amdgcn-amd-amdhsa--gfx950"):
    fat = os.path.join(out_dir, "fatbin.bin")
    subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", lib, fat], check=True)
    blob = open(fat, "rb").read()
    starts = [m.start() for m in re.finditer(rb"CCOB|__CLANG_OFFLOAD_BUNDLE__", blob)]
    paths = []
    for i, at in enumerate(starts):
        part = blob[at:starts[i + 1] if i + 1 < len(starts) else len(blob)]
        if part[:4] == b"CCOB":                                  # compressed bundle: magic, u16 version, u16 method, total size (u32 in v2, u64 in v3), ...
            version = struct.unpack_from("<H", part, 4)[0]
            if version >= 3:
                part = part[:struct.unpack_from("<Q", part, 8)[0]]
            elif version == 2:
                part = part[:struct.unpack_from("<I", part, 8)[0]]
        src, dst = os.path.join(out_dir, "bundle%d.bin" % i), os.path.join(out_dir, "code%d.elf" % i)
        with open(src, "wb") as fh:
            fh.write(part)
        subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--targets=" + target, "--input=" + src, "--output=" + dst], check=True)
        paths.append(dst)
    return paths
