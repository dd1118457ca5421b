"""Static scan of the kernels' ISA for the LDS pattern that round 6 found to be unsafe beside another kernel's ds_read_b64_tr_b16:
a COUNTED `s_waitcnt lgkmcnt(N > 0)` while LDS reads of BOTH banking classes are in flight (4-byte class: ds_read_b32 / ds_read2_b32 /
u8 / u16; 8/16-byte class: ds_read_b64 / b96 / b128 / ds_read2_b64 / tr reads).  mt_token_mha_fwd's value sweep
([6 x b128, 4 x read2_b32, 2 x b128] + lgkmcnt(5)) returned stale 16-byte results in lanes 48-63 once in ~10 launches beside
mt_gemm_tn_f16 on another stream; the same instructions behind ONE lgkmcnt(0) never did (tools/diag/victim_stress2.py).
    for f in modaltune_amd/csrc/*.hip; do hipcc --offload-arch=gfx950 -O3 -S --cuda-device-only -o /tmp/isa/$(basename $f .hip).s $f; done
    python tools/diag/lds_wait_scan.py /tmp/isa/*.s"""
import re
import sys

SMALL = re.compile(r"\bds_read(2|2st64)?_(b32|u8|i8|u16|i16|u8_d16|u16_d16)\b|\bds_read_(u8|i8|u16|i16)\b")
WIDE = re.compile(r"\bds_read(2|2st64)?_(b64|b96|b128)\b|\bds_read_b64_tr_b\d+\b|\bds_read_b96_tr_b6\b")
ANYLDS = re.compile(r"^\s*ds_")
WAIT = re.compile(r"s_waitcnt\b(.*)")

def scan(path):
    """{kernel symbol: number of counted lgkmcnt waits taken while LDS reads of both banking classes were in flight}"""
    kernel, inflight, flagged = None, [], {}
    for line in open(path):
        m = re.match(r"^(_Z\w+|mt_\w+|\w+_kernel\w*):", line)
        if m:
            kernel, inflight = m.group(1), []
            continue
        if kernel is None:
            continue
        if "s_endpgm" in line:
            kernel = None
            continue
        s = line.split(";")[0]
        if ANYLDS.match(s):
            inflight.append("small" if SMALL.search(s) else "wide" if WIDE.search(s) else "other")
            continue
        w = WAIT.search(s)
        if w:
            mm = re.search(r"lgkmcnt\((\d+)\)", w.group(1))
            if mm is None:
                continue          # (a wait that names only vmcnt leaves the LDS counter alone)
            n = int(mm.group(1))
            if n > 0 and {"small", "wide"} <= set(inflight):
                flagged[kernel] = flagged.get(kernel, 0) + 1
            # what is in flight stays in flight across basic blocks (conservative); a wait retires all but the n youngest
            inflight = inflight[len(inflight) - n:] if 0 < n < len(inflight) else ([] if n == 0 else inflight)
    return flagged


if __name__ == "__main__":
    for path in sys.argv[1:]:
        for k, c in sorted(scan(path).items()):
            print(f"{path.split('/')[-1]:18s} {c:4d} counted waits over mixed-class LDS reads in {k}")
