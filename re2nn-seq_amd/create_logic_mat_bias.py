"""Priority matrices per dataset (reference src_seq/create_logic_mat_bias.py): identity with -1
at [i-x][b-x] so that a B- score suppresses the matching I- score, plus dataset-specific pairs."""
import numpy as np

_EXTRA = {
    'MITM': [('o', 'i-year'), ('o', 'b-actor')],
    'SNIPS': [('b-playlist_owner', 'b-playlist')],
}


def _bio_priority(s2i):
    mat = np.eye(len(s2i))
    for slot, idx in s2i.items():
        if 'b-' in slot:
            inside = 'i-' + slot[2:]
            if inside in s2i:
                mat[s2i[inside]][idx] = -1
    return mat


def create_mat_priority_MITR(s2i):
    return np.eye(len(s2i))


def create_mat_priority(s2i, args):
    name = args.dataset
    if 'MITM' in name:
        key = 'MITM'
    elif 'MITR' in name:
        return create_mat_priority_MITR(s2i)
    elif 'ATIS' in name:
        key = 'ATIS'
    elif 'SNIPS' in name:
        key = 'SNIPS'
    else:
        raise NotImplementedError(name)
    mat = _bio_priority(s2i)
    for a, b in _EXTRA.get(key, []):
        mat[s2i[a]][s2i[b]] = -1
    return mat
