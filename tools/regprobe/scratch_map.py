"""Where do the scratch (spill) instructions of each kernel sit?  tools/regprobe/scratch_map.py /tmp/probe.s
Prints per kernel the line offsets of scratch ops, barriers and s_endpgm inside its body, so that a spill in the general (deferred) path can be told
from one in the hot path (which comes after the chunk prologue and between the barriers)."""
import re
import sys
lines = open(sys.argv[1]).read().split('\n')
starts = [(i, l.split(':')[0]) for i, l in enumerate(lines) if re.match(r'^_Z\w+:', l)]
starts.append((len(lines), 'END'))
for (a, name), (b, _) in zip(starts, starts[1:]):
    body = lines[a:b]
    sc = [i for i, l in enumerate(body) if 'scratch_' in l]
    if not sc:
        continue
    ends = [i for i, l in enumerate(body) if 's_endpgm' in l]
    bar = [i for i, l in enumerate(body) if 's_barrier' in l]
    print(name[8:60], 'lines', len(body), '| scratch ops', len(sc), 'at', sc[0], '..', sc[-1], '| endpgm', ends, '| barriers', bar)
    # histogram by 10 % of the body
    hist = [0] * 10
    for i in sc:
        hist[min(9, i * 10 // len(body))] += 1
    print('    scratch per decile of the body:', hist)
