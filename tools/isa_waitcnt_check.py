"""For every MFMA kernel in an ISA dump (hipcc -S --cuda-device-only): does the main loop wait for its global loads
(s_waitcnt vmcnt) BEFORE the MFMA block of the same iteration?  (That exposes the full memory latency every K-step.)
    python tools/isa_waitcnt_check.py /tmp/isa/conv.s"""
import re
import sys

lines = open(sys.argv[1]).read().split('\n')
starts = [i for i, l in enumerate(lines) if re.match(r'^_Z\d+conv_\w+.*:\s', l)]
for s in starts:
    name = lines[s].split(':')[0]
    end = next(i for i in range(s, len(lines)) if 's_endpgm' in lines[i])
    body = [re.sub(r'\s+;.*$', '', l.strip()) for l in lines[s:end]]
    body = [l for l in body if l and not l.startswith(';')]
    mf = [i for i, l in enumerate(body) if l.startswith('v_mfma')]
    if not mf:
        continue
    # main-loop MFMA run = the largest group of MFMAs; look back from its first MFMA to the previous label / barrier
    first = mf[0]
    j = first
    waits = []
    while j > 0 and not body[j].startswith('s_barrier'):
        if 'vmcnt' in body[j]:
            waits.append(body[j])
        j -= 1
    short = re.sub(r'^_Z\d+', '', name)
    print("%-62s %s" % (short[:62], "WAITS BEFORE MFMA: " + "; ".join(waits) if waits else "ok"))
