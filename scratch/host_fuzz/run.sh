#!/bin/bash
# AddressSanitizer + UBSan runs of libbsr_host.so's two sources on the CPU (sanitizers are a CPU-build matter on this pool): 20 000 damaged /
# truncated zlib streams with right and wrong output sizes through bsr_inflate_zlib, 20 000 random scanline sets (every filter type, the
# undefined ones included, 1-4 bytes per pixel) through bsr_png_unfilter.   bash scratch/host_fuzz/run.sh
set -e
cd "$(dirname "$0")"
python3 - <<'P'
import glob, struct, zlib
b = open(sorted(glob.glob("../../tests/golden/UCB/train/input/*/*.png"))[0], "rb").read()
o, parts = 8, []
while o + 12 <= len(b):
    m, = struct.unpack(">I", b[o:o + 4])
    if b[o + 4:o + 8] == b"IDAT":
        parts.append(b[o + 8:o + 8 + m])
    o += 12 + m
z = b"".join(parts)
open("/tmp/bsr_fuzz_stream.bin", "wb").write(z)
open("/tmp/bsr_fuzz_stream.len", "w").write(str(len(zlib.decompress(z))))
P
gcc -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -o /tmp/bsr_inflate_fuzz inflate_fuzz.c
gcc -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -o /tmp/bsr_unfilter_fuzz unfilter_fuzz.c
/tmp/bsr_inflate_fuzz /tmp/bsr_fuzz_stream.bin "$(cat /tmp/bsr_fuzz_stream.len)"
/tmp/bsr_unfilter_fuzz
