"""Shared helpers for the parity tests."""
import numpy as np


def seeds_from_chains(batch_in, chains, only_ok=True):
    """hlala_seeds_in dict from a batch and its stage-A chain output (status == 0 chains only)."""
    stride = chains["_stride"]
    n_reads = 2 * batch_in["n_pairs"]
    chain_read = np.zeros(batch_in["n_chains"], np.int32)
    for r in range(n_reads):
        chain_read[batch_in["chain_off"][r]:batch_in["chain_off"][r + 1]] = r
    keep = [c for c in range(batch_in["n_chains"]) if (chains["status"][c] == 0 or not only_ok)]
    col_off = [0]
    lv, ed, g, s = [], [], [], []
    for c in keep:
        n = int(chains["n_cols"][c]); b = c * stride
        lv.append(chains["col_level"][b:b + n]); ed.append(chains["col_edge"][b:b + n])
        g.append(chains["col_gchar"][b:b + n]); s.append(chains["col_schar"][b:b + n])
        col_off.append(col_off[-1] + n)
    cat = lambda xs, dt: (np.concatenate(xs).astype(dt) if xs else np.zeros(0, dt))
    return dict(n_reads=n_reads, read_off=batch_in["read_off"], read_bases=batch_in["read_bases"],
                read_quals=batch_in["read_quals"], n_chains=len(keep),
                chain_read=chain_read[keep].astype(np.int32),
                chain_seq_begin=chains["seq_begin"][keep].astype(np.int32),
                chain_seq_end=chains["seq_end"][keep].astype(np.int32),
                chain_reverse=np.asarray(batch_in["chain_reverse"])[keep].astype(np.uint8),
                col_off=np.asarray(col_off, np.int32), col_level=cat(lv, np.int32), col_edge=cat(ed, np.int32),
                col_gchar=cat(g, np.uint8), col_schar=cat(s, np.uint8), _keep=np.asarray(keep, np.int32))


def chain_cols(d, c):
    st = d["_stride"]; n = int(d["n_cols"][c]); b = c * st
    return (d["col_level"][b:b + n], d["col_edge"][b:b + n], bytes(d["col_gchar"][b:b + n]),
            bytes(d["col_schar"][b:b + n]), d["col_fromseed"][b:b + n])


def compare_chains(got, exp, n_chains, check_ll=True, check_dp=True, ll_rtol=1e-12, label=""):
    """Bit-exact comparison of chain-level outputs (integers/bytes) and LL within ll_rtol."""
    bad = []
    for c in range(n_chains):
        if got["status"][c] != exp["status"][c]:
            bad.append((c, "status", int(got["status"][c]), int(exp["status"][c]))); continue
        if exp["status"][c] != 0:
            continue
        if got["n_cols"][c] != exp["n_cols"][c]:
            bad.append((c, "n_cols", int(got["n_cols"][c]), int(exp["n_cols"][c]))); continue
        for k in ("seq_begin", "seq_end"):
            if got[k][c] != exp[k][c]:
                bad.append((c, k, int(got[k][c]), int(exp[k][c])))
        a, b = chain_cols(got, c), chain_cols(exp, c)
        for i, name in enumerate(("level", "edge", "gchar", "schar", "fromseed")):
            same = (a[i] == b[i]) if isinstance(a[i], bytes) else np.array_equal(a[i], b[i])
            if not same:
                bad.append((c, name, a[i], b[i]))
        if check_dp:
            for k in ("dp_iters", "dp_score"):
                if not np.array_equal(got[k][2 * c:2 * c + 2], exp[k][2 * c:2 * c + 2]):
                    bad.append((c, k, got[k][2 * c:2 * c + 2].tolist(), exp[k][2 * c:2 * c + 2].tolist()))
        if check_ll:
            e = exp["ll"][c]
            if not (abs(got["ll"][c] - e) <= ll_rtol * max(1.0, abs(e))):
                bad.append((c, "ll", float(got["ll"][c]), float(e)))
    assert not bad, f"{label}: {len(bad)} chain mismatches, first: {bad[:3]}"
