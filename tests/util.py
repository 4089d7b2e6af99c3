"""Shared helpers for the parity tests (oracle = checker; papc_amd = product)."""
import numpy as np


def seeded_weights(chans, seed, bias_scale=0.1):
    """[(conv_w [Cout,Cin], conv_b, gamma, beta)] with non-trivial gamma/beta so BN folding is exercised
    (including NEGATIVE gammas: the max must then pick the smallest pre-BN value)."""
    rng = np.random.default_rng(seed)
    ws = []
    for cin, cout in zip(chans[:-1], chans[1:]):
        w = (rng.normal(size=(cout, cin)) * np.sqrt(2.0 / cin)).astype(np.float32)
        b = (rng.normal(size=cout) * bias_scale).astype(np.float32)
        g = rng.uniform(0.5, 1.5, size=cout).astype(np.float32) * rng.choice([1.0, 1.0, 1.0, -1.0], size=cout).astype(np.float32)
        bt = (rng.normal(size=cout) * 0.2).astype(np.float32)
        ws.append((w, b, g, bt))
    return ws


def assert_close(got, ref, rel=1e-5, what=""):
    """|got-ref| <= rel * max|ref| elementwise-max criterion (activations are O(1) after BN); the north-star
    tolerance for MLP activations is 1e-5 relative."""
    got = np.asarray(got, np.float64)
    ref = np.asarray(ref, np.float64)
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    scale = max(float(np.max(np.abs(ref))), 1e-30)
    err = float(np.max(np.abs(got - ref))) / scale
    assert err <= rel, "%s: max rel err %.3e > %.1e" % (what, err, rel)
    return err
