"""Reverses the predecessor order (= the use-list order of the block) of two-predecessor blocks of the kernel in an .ll file
with `uselistorder_bb` directives (numbered blocks cannot be named in the directive, so they get names first).
usage: reverse_preds.py IN.ll OUT.ll all|N,N,...     (all two-predecessor blocks, or the numbered blocks given)"""
import re, sys
src, dst, mode = sys.argv[1:4]
s = open(src).read()
fn = re.search(r'@_Z7k_merge\w+', s).group(0)
blocks = [b for b, p, q in re.findall(r'^(\d+):\s+; preds = %(\d+), %(\d+)$', s, flags=re.M)]
if mode != "all":
    want = mode.split(",")
    assert all(w in blocks for w in want), (want, blocks)
    blocks = want
names = []
for b in blocks:
    s = re.sub(r'^%s:' % b, 'bb%s:' % b, s, flags=re.M)
    s = re.sub(r'%%%s\b' % b, '%%bb%s' % b, s)
    names.append('bb' + b)
s += '\n' + ''.join('uselistorder_bb %s, %%%s, { 1, 0 }\n' % (fn, n) for n in names)
open(dst, 'w').write(s)
print(len(names), 'blocks reversed:', ' '.join(names))
