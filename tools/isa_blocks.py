"""Basic-block listing of one kernel from a `-gline-tables-only -save-temps` assembly file, attributed to the phases of substep().

    python tools/isa_blocks.py /tmp/g/bp_capi-hip-amdgcn-amd-amdhsa-gfx950.s _Z20k_physics_step_sched9 > blocks.txt

One line per basic block: label, loop depth (from the compiler's block comments), phase of substep() that most of its instructions
belong to (outermost bp_physics.hpp line of the inline chain), instruction counts by class, range of bp_physics.hpp lines.
"""
import collections
import re
import sys

# phases of substep() as line ranges of csrc/bp_physics.hpp (update when the file moves; `grep -n "// ---- " csrc/bp_physics.hpp`)
PH = [(291, 307, '0head'), (308, 381, '1integrate'), (382, 395, '2refresh'), (396, 480, '3candidates'), (481, 530, '4a_cached_planes'),
      (531, 689, '4a_bound_rounds+search'), (690, 861, '4b_manifold'), (862, 983, '4c_deliver'), (984, 1000, '5events+filter'),
      (1001, 1038, '6a_prestep'), (1039, 1066, '6a_warmset'), (1067, 1106, '6a_colour'), (1107, 1132, '6b_velint'), (1133, 1156, '6c_warmstart'),
      (1157, 1282, '6d_solver'), (1283, 1337, '7post'), (1338, 1404, '7mvlist'), (256, 280, 'support_queries'), (216, 238, 'world_from_pose'),
      (163, 213, 'refresh_body')]


def phase_of(chain):
    ph = [int(m.group(1)) for m in re.finditer(r'bp_physics\.hpp:(\d+)', chain)]
    if ph:
        l = ph[-1]
        for a, b, n in PH:
            if a <= l <= b:
                return n, l
        return 'phys_helper', l
    k = [int(m.group(1)) for m in re.finditer(r'bp_kernels\.hpp:(\d+)', chain)]
    return ('kernels', k[0] if k else 0)


def classify(op):
    if op.startswith('v_readlane') or op.startswith('v_writelane'):
        return 'rl'
    if op.startswith('v_'):
        return 'valu'
    if op.startswith('s_waitcnt') or op.startswith('s_nop'):
        return 'wait'
    if op.startswith('s_cbranch') or op.startswith('s_branch'):
        return 'br'
    if op.startswith('s_load'):
        return 'smem'
    if op.startswith('s_'):
        return 'salu'
    if op.startswith('ds_'):
        return 'lds'
    if op.startswith(('global_', 'flat_', 'buffer_', 'scratch_')):
        return 'vmem'
    return 'other'


def main():
    path, kernel = sys.argv[1], sys.argv[2]
    inside = False
    blocks = []
    cur = None
    curp = ('?', 0)
    for raw in open(path):
        s = raw.strip()
        if not inside:
            if raw.startswith(kernel):
                inside = True
                cur = {'label': 'entry', 'depth': 0, 'cnt': collections.Counter(), 'ph': collections.Counter(), 'lines': []}
                blocks.append(cur)
            continue
        if s.startswith('.Lfunc_end'):
            break
        if s.startswith('.loc'):
            curp = phase_of(s)
            continue
        m = re.match(r'^(\.LBB\S+):\s*(;.*)?$', s)
        if m:
            d = re.search(r'Depth=(\d+)', m.group(2) or '')
            cur = {'label': m.group(1), 'depth': int(d.group(1)) if d else 0, 'cnt': collections.Counter(), 'ph': collections.Counter(), 'lines': []}
            blocks.append(cur)
            continue
        if not s or s.startswith('.') or s.startswith(';'):
            continue
        op = s.split()[0]
        if re.match(r'^[a-z_0-9]+$', op):
            cur['cnt'][classify(op)] += 1
            cur['ph'][curp[0]] += 1
            if curp[0] not in ('kernels',):
                cur['lines'].append(curp[1])
    tot = collections.defaultdict(collections.Counter)
    for b in blocks:
        n = sum(b['cnt'].values())
        if n == 0:
            continue
        p = b['ph'].most_common(1)[0][0]
        tot[p].update(b['cnt'])
        ls = b['lines']
        print('%-12s d%d %-13s n=%3d %s  L%s-%s' % (b['label'], b['depth'], p, n, ' '.join('%s=%d' % kv for kv in sorted(b['cnt'].items())),
                                                  min(ls) if ls else '', max(ls) if ls else ''))
    print()
    for p in sorted(tot):
        print('%-13s %4d  %s' % (p, sum(tot[p].values()), dict(tot[p])))


if __name__ == '__main__':
    main()
