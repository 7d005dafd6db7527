"""CPU: the oracle still reproduces the committed fixture (guards against silent drift of the checker)."""
from golden_util import check_against_golden, load_golden


def test_oracle_reproduces_golden(oracle):
    g, c, b, e = load_golden()
    r = oracle(g, c, insert_mean=b["insert_mean"], insert_sd=b["insert_sd"], rng_seed=4242, max_columns=384).align_batch(b)
    check_against_golden(r["ext"], r["pairs"], e, 384)
