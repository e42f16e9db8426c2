"""One process per GPU: keep a rank's host thread on the cores of ITS GPU's NUMA node.

A rank of the self-play engine is a single host thread that enqueues hipGraphs and reads a log from pinned memory (rlzero_amd.selfplay);
on an 8-GPU node the GPUs hang off two sockets, and a rank scheduled on the far socket pays the inter-socket hop on every enqueue
and every pinned read.  ``pin_to_gpu`` sets the calling process's CPU affinity from sysfs alone -- no numactl / taskset hop (a
process that has touched the GPU must not exec), no GPU call (it runs BEFORE the HIP runtime starts) -- and returns what it did
for the benchmark line.  Ranks that share a NUMA node split its cores evenly, so eight host threads never pile onto the same cores.

Device order: HIP enumerates the GPUs in KFD topology order (``/sys/class/kfd/kfd/topology/nodes/<n>``, the nodes with SIMDs), after
``HIP_VISIBLE_DEVICES`` / ``ROCR_VISIBLE_DEVICES`` if they are plain index lists; a node's ``drm_render_minor`` leads to its PCI
device and ``numa_node`` / ``local_cpulist`` there.
"""
import os
import re


def _read(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


def parse_cpulist(text):
    """'0-3,8,10-11' -> [0, 1, 2, 3, 8, 10, 11]."""
    cpus = []
    for part in (text or '').split(','):
        part = part.strip()
        if not part:
            continue
        if '-' in part:
            lo, hi = part.split('-', 1)
            cpus.extend(range(int(lo), int(hi) + 1))
        else:
            cpus.append(int(part))
    return cpus


def format_cpulist(cpus):
    cpus = sorted(set(cpus))
    out, i = [], 0
    while i < len(cpus):
        j = i
        while j + 1 < len(cpus) and cpus[j + 1] == cpus[j] + 1:
            j += 1
        out.append('%d-%d' % (cpus[i], cpus[j]) if j > i else '%d' % cpus[i])
        i = j + 1
    return ','.join(out)


def gpu_numa_nodes(sysfs='/sys'):
    """-> [(numa node, [cpus local to it]) per GPU in KFD topology order]; numa node -1 / no cpus when sysfs does not say."""
    base = os.path.join(sysfs, 'class', 'kfd', 'kfd', 'topology', 'nodes')
    try:
        nodes = sorted((int(n) for n in os.listdir(base) if n.isdigit()))
    except OSError:
        return []
    out = []
    for n in nodes:
        props = _read(os.path.join(base, str(n), 'properties')) or ''
        kv = dict(re.findall(r'^(\w+)\s+(\d+)$', props, flags=re.M))
        if int(kv.get('simd_count', '0')) == 0:
            continue   # a CPU node
        minor = kv.get('drm_render_minor')
        dev = os.path.join(sysfs, 'class', 'drm', 'renderD%s' % minor, 'device') if minor is not None else None
        numa = _read(os.path.join(dev, 'numa_node')) if dev else None
        cpus = parse_cpulist(_read(os.path.join(dev, 'local_cpulist'))) if dev else []
        numa = int(numa) if numa is not None and re.fullmatch(r'-?\d+', numa) else -1
        if not cpus and numa >= 0:
            cpus = parse_cpulist(_read(os.path.join(sysfs, 'devices', 'system', 'node', 'node%d' % numa, 'cpulist')))
        out.append((numa, cpus))
    return out


def _visible(n_gpus, env):
    """Indices of the visible GPUs in device-ordinal order (plain index lists only; anything else: the identity)."""
    order = list(range(n_gpus))
    # ROCR_VISIBLE_DEVICES filters first (the runtime below HIP), then exactly ONE of HIP_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES: in
    # HIP the latter is an alias read only when the former is unset, not a further filter
    hip_var = 'HIP_VISIBLE_DEVICES' if env.get('HIP_VISIBLE_DEVICES') not in (None, '') else 'CUDA_VISIBLE_DEVICES'
    for var in ('ROCR_VISIBLE_DEVICES', hip_var):
        val = env.get(var)
        if val is None or val == '':
            continue
        try:
            pick = [int(x) for x in val.split(',') if x.strip() != '']
        except ValueError:
            return order
        if any(i < 0 or i >= len(order) for i in pick):
            return order
        order = [order[i] for i in pick]
    return order


def plan(device_ordinal, local_ranks, allowed, sysfs='/sys', env=None):
    """The cores for the rank on ``device_ordinal`` when ``local_ranks`` (device ordinals of ALL ranks of this node) run together
    and the process may use ``allowed`` -> dict(numa_node, cpus) or None when nothing is known (no pinning then)."""
    env = os.environ if env is None else env
    gpus = gpu_numa_nodes(sysfs)
    if not gpus:
        return None
    order = _visible(len(gpus), env)
    if device_ordinal >= len(order):
        return None
    numa, cpus = gpus[order[device_ordinal]]
    cpus = [c for c in cpus if c in set(allowed)]
    if not cpus:
        return None
    # ranks whose GPUs share this NUMA node split its cores evenly (in device order)
    sharers = sorted(o for o in set(local_ranks) if o < len(order) and gpus[order[o]][0] == numa)
    if device_ordinal in sharers and len(sharers) > 1 and len(cpus) >= len(sharers):
        k, n = sharers.index(device_ordinal), len(sharers)
        share = len(cpus) // n
        cpus = cpus[k * share:(k + 1) * share] if k < n - 1 else cpus[k * share:]
    return {'numa_node': numa, 'cpus': cpus}


def pin_to_gpu(device_ordinal, local_world=1, sysfs='/sys', env=None, apply=True):
    """Pin the calling process to the cores near GPU ``device_ordinal`` (one of ``local_world`` ranks on devices 0 .. local_world-1
    of this node).  Call before the first GPU call.  -> dict for the benchmark line: numa_node, cpus (a cpulist string), pinned."""
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except AttributeError:
        return {'pinned': False, 'why': 'no sched_getaffinity on this platform'}
    p = plan(device_ordinal, list(range(max(1, int(local_world)))), allowed, sysfs, env)
    if p is None:
        return {'pinned': False, 'why': 'sysfs names no NUMA node / local cores for this GPU', 'cpus': format_cpulist(allowed)}
    if apply:
        os.sched_setaffinity(0, p['cpus'])
    return {'pinned': bool(apply), 'numa_node': p['numa_node'], 'cpus': format_cpulist(p['cpus']), 'n_cpus': len(p['cpus'])}
