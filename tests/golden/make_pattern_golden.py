"""Golden fixture for the codebook pattern providers (reference audiocraft/modules/codebooks_patterns.py:305-552 and the
`Pattern` maps :116-269): layouts and index maps recorded from the unmodified reference for every provider of its builder
(`delay` incl. flatten_first / empty_initial, `parallel`, `unroll`, `coarse_first`, `musiclm`) at a few geometries.

Run in the build container only:   python tests/golden/make_pattern_golden.py   ->  tests/golden/patterns.json
"""
import json
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, '..', '..'))
from oracle import refstubs  # noqa: E402,F401
import audiocraft.modules.codebooks_patterns as R  # noqa: E402

PROVIDERS = {'delay': 'DelayedPatternProvider', 'parallel': 'ParallelPatternProvider', 'unroll': 'UnrolledPatternProvider',
             'coarse_first': 'CoarseFirstPattern', 'musiclm': 'MusicLMPattern'}


def cases():
    out = []
    for n_q, T in [(4, 7), (3, 4), (2, 1), (4, 2), (6, 5)]:
        out += [('delay', dict(delays=list(range(n_q))), n_q, T), ('delay', dict(delays=[0] * n_q, flatten_first=2), n_q, T),
                ('delay', dict(delays=list(range(n_q)), flatten_first=3, empty_initial=2), n_q, T),
                ('delay', dict(empty_initial=1), n_q, T), ('delay', dict(delays=[0, 0] + [2] * (n_q - 2)), n_q, T),
                ('delay', dict(delays=[1] + [3] * (n_q - 1)), n_q, T),     # no codebook without delay: max_delay < max(delays)
                ('parallel', dict(), n_q, T), ('parallel', dict(empty_initial=2), n_q, T), ('unroll', dict(), n_q, T),
                ('coarse_first', dict(), n_q, T), ('coarse_first', dict(delays=list(range(n_q - 1))), n_q, T)]
        if n_q % 2 == 0:
            out.append(('musiclm', dict(group_by=2), n_q, T))
        if n_q >= 3:
            out += [('unroll', dict(flattening=[0] + [1] * (n_q - 1)), n_q, T),
                    ('unroll', dict(flattening=[0] + [1] * (n_q - 1), delays=[0] + [3] * (n_q - 1)), n_q, T),
                    ('unroll', dict(flattening=[0, 1] + [3] * (n_q - 2), delays=[0, 0] + [2] * (n_q - 2)), n_q, T)]
    return out


if __name__ == '__main__':
    g = torch.Generator().manual_seed(0)
    recs = []
    for name, kw, n_q, T in cases():
        p = getattr(R, PROVIDERS[name])(n_q, **kw).get_pattern(T)
        rec = dict(provider=name, kwargs=kw, n_q=n_q, timesteps=T, layout=[[[c.t, c.q] for c in s] for s in p.layout],
                   num_sequence_steps=p.num_sequence_steps, max_delay=p.max_delay, valid_steps=len(p.valid_layout), maps=[],
                   first_steps=[[p.get_first_step_with_timesteps(t, q) for q in [None] + list(range(n_q))] for t in range(T + 1)])
        z = torch.randint(0, 50, (2, n_q, T), generator=g)
        rec['z'] = z.tolist()
        for keep in (False, True):
            v, i, m = p.build_pattern_sequence(z, 99, keep)
            m_rec = dict(keep=keep, values=v.tolist(), indexes=i.tolist(), mask=m.int().tolist(), revert=[])
            for S in sorted({v.shape[-1], max(v.shape[-1] - 2, 1), 1}):
                if S > v.shape[-1]:
                    continue
                s = torch.randint(0, 50, (2, n_q, S), generator=g)
                rv, ri, rm = p.revert_pattern_sequence(s, -1, keep)
                _, li, lm = p.revert_pattern_logits(torch.zeros(1, 1, n_q, S), 0., keep)
                m_rec['revert'].append(dict(S=S, s=s.tolist(), values=rv.tolist(), indexes=ri.tolist(), mask=rm.int().tolist(),
                                            logits_indexes=li.tolist(), logits_mask=lm.int().tolist()))
            rec['maps'].append(m_rec)
        recs.append(rec)
    path = os.path.join(HERE, 'patterns.json')
    json.dump(recs, open(path, 'w'), separators=(',', ':'))
    print(f'wrote {path}: {len(recs)} cases, {os.path.getsize(path) / 1024:.1f} KiB')
