"""Per-kernel register / scratch / LDS usage from a `hipcc -Rpass-analysis=kernel-resource-usage` log (stderr of a compile)."""
import re, sys
txt = open(sys.argv[1]).read()
for b in txt.split('Function Name: ')[1:]:
    name = b.split('\n')[0].strip()
    def g(k):
        m = re.search(re.escape(k) + r': (\d+)', b)
        return int(m.group(1)) if m else -1
    m = re.search(r'(eh_\w+?)I((?:L[ib]\d+E)+)', name)
    tag = name[:70]
    if m:
        tag = m.group(1) + '<' + ','.join(re.findall(r'L[ib](\d+)E', m.group(2))) + '>'
    print(f"{tag}: VGPR {g('VGPRs')} AGPR {g('AGPRs')} scratch {g('ScratchSize [bytes/lane]')} spillV {g('VGPRs Spill')} "
          f"spillS {g('SGPRs Spill')} occ {g('Occupancy [waves/SIMD]')} LDS {g('LDS Size [bytes/block]')}")
