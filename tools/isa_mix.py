"""Instruction mix of a built kernel (measurement aid, CPU only: hipcc cross-compiles).

    python tools/isa_mix.py tbk_eig_band.hip band_reduce_kernelILi256ELi1ELb1ELi0E [extra hipcc flags]

Prints registers / scratch of every kernel of the source file, then -- for the kernel whose mangled name contains the
second argument -- the instruction classes of every basic block of more than 25 instructions and an opcode histogram.
What the vector unit issues beside the matrix instructions is matrix-pipe time on gfx950."""
import collections, os, re, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src_name, want = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else None)
out = '/tmp/isa_mix_%s.s' % os.path.basename(src_name)
subprocess.run(['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '--offload-arch=gfx950', '-I' + ROOT + '/include', '-I/opt/rocm/include',
                '--cuda-device-only', '-S', os.path.join(ROOT, 'tbmodels_amd', 'csrc', src_name), '-o', out] + sys.argv[3:],
               check=True, stderr=subprocess.DEVNULL)
src = open(out).read().split('\n')
name = None
for l in src:
    m = re.match(r'\s*\.set (\S+?)\.(num_vgpr|num_agpr|private_seg_size), (\d+)', l)
    if not m:
        continue
    if m.group(2) == 'num_vgpr':
        name, vg = m.group(1), m.group(3)
    elif m.group(2) == 'private_seg_size' and name == m.group(1):
        print('%-90s vgpr %3s scratch %s' % (re.sub(r'^_ZN12_GLOBAL__N_1\d+', '', name)[:90], vg, m.group(3)))
if want:
    start = [i for i, l in enumerate(src) if re.match(r'^_Z\S*' + re.escape(want) + r'\S*:', l)][0]
    end = [i for i, l in enumerate(src) if i > start and '.amdhsa_kernel' in l][0]
    blk, blocks, tot, ops = None, collections.OrderedDict(), collections.Counter(), collections.Counter()
    for l in src[start:end]:
        m = re.match(r'^(\.LBB\d+_\d+):', l)
        if m:
            blk = m.group(1)
            blocks[blk] = collections.Counter()
            continue
        s = l.strip()
        if blk is None or not s or s[0] in ';.':
            continue
        op = s.split()[0]
        k = ('mfma' if op.startswith('v_mfma') else 'valu' if op.startswith('v_') else 'salu' if op.startswith('s_') else
             'lds' if op.startswith('ds_') else 'vmem' if op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')) else 'other')
        blocks[blk][k] += 1
        tot[k] += 1
        ops[op] += 1
        if op.startswith('v_cndmask'):
            blocks[blk]['cndmask'] += 1
    for b, c in blocks.items():
        if sum(c.values()) > 25:
            print(b, dict(c))
    print(want, dict(tot))
    print(', '.join('%s %d' % kv for kv in ops.most_common(24)))
