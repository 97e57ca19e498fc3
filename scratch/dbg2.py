import sys, numpy as np, torch
sys.path.insert(0,'.')
from tests.golden_util import load_golden, split_ids, split_outputs
from tests.test_hip_decode_parity import _replay_decoding
for name in sys.argv[1:]:
    g = load_golden(name)
    try:
        ids_log, outs, bank = _replay_decoding(g, 0)
    except Exception as e:
        print(name, 'EXC', repr(e)); continue
    ref_out = split_outputs(g)[1:]
    errs = [float((a-b).abs().max()) for a,b in zip(outs, ref_out)]
    bad = [i for i,e in enumerate(errs) if e > 1e-3]
    print(name, 'n', len(errs), 'first bad out step', bad[:3], 'err', [round(errs[i],4) for i in bad[:3]])
    if g['meta']['config']['kv_policy'] in ('roco','h2o_head','tova'):
        ref = split_ids(g)
        for s,(a,b) in enumerate(zip(ids_log, ref)):
            if not np.array_equal(a,b):
                print('  first id mismatch at evict step', s, 'ours', a.reshape(-1)[:8], 'ref', b.reshape(-1)[:8]); break
        else: print('  ids all equal', len(ids_log), len(ref))
