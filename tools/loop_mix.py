"""instruction mix of the k-loop of every gemm_kernel instantiation in an assembly listing (hipcc -S --cuda-device-only)"""
import re, sys
s = open(sys.argv[1] if len(sys.argv) > 1 else '/tmp/gemm.s').read()
starts = [(m.start(), m.group(1)) for m in re.finditer(r'\n(_ZN3vbg11gemm_kernelILi\d+ELi\d+ELi\d+ELi\d+ELi\d+ELi\d+ELb\dEEEv13vbg_gemm_desc):', s)]
starts.append((len(s), 'end'))
for (a, name), (b, _) in zip(starts, starts[1:]):
    g = re.match(r'_ZN3vbg11gemm_kernelILi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELb(\d)', name).groups()
    f = s[a:b]
    blocks, cur, lab = [], [], 'entry'
    for l in f.split('\n'):
        if re.match(r'^\.LBB\d+_\d+:', l):
            blocks.append((lab, cur)); cur = []; lab = l.strip()
        else:
            t = l.strip()
            if t and not t.startswith(('.', ';', '//')) and not t.endswith(':'):
                cur.append(t)
    blocks.append((lab, cur))
    # blocks that belong to the loop containing the MFMAs: from the first to the last block with mfma, widened to loop markers
    idx = [i for i, (lab, ins) in enumerate(blocks) if any('v_mfma' in x for x in ins)]
    if not idx:
        continue
    lo, hi = idx[0], idx[-1]
    ins = [x for lab, bl in blocks[lo:hi + 1] for x in bl]
    cnt = lambda p: sum(1 for x in ins if re.match(p, x))
    vg = re.search(r'\.vgpr_count:\s+(\d+)', f) or re.search(r'; NumVgprs: (\d+)', f)
    print('x'.join(g[:3]), 'A%s B%s vec%s' % (g[4], g[5], g[6]), 'blocks', hi - lo + 1, 'mfma', cnt(r'v_mfma'), 'valu', cnt(r'v_(?!mfma)'),
          'salu', cnt(r's_(?!waitcnt|barrier|nop|cbranch|branch)'), 'ds', cnt(r'ds_'), 'vmem', cnt(r'(global|buffer)_'), 'vgprs', vg.group(1) if vg else '?')
