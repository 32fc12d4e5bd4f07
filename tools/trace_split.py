# per-level timing of a split / replace on random keys (timing only; run with TFHE_HIP_HOST_TRACE=1)
import sys, time, numpy as np
sys.path.insert(0, '.')
import tfhe_rs_string_amd as T
p = T.PARAM_MESSAGE_2_CARRY_2_KS_PBS()
rng = np.random.default_rng(7)
ctx = T.Context(0)
sk = T.ServerKey(ctx, p, rng.integers(0, 2**64, p.bsk_len, dtype=np.uint64), rng.integers(0, 2**64, p.ksk_len, dtype=np.uint64))
key = T.StringServerKey(sk.handle, p)
L = int(sys.argv[1]) if len(sys.argv) > 1 else 52
s = rng.integers(0, 2**64, (L, 4, p.big_size), dtype=np.uint64)
for name, fn in [("split ', '", lambda: key.split(s, ", ")), ("split ','", lambda: key.split(s, ",")),
                 ("split enc 2", lambda: key.split(s, s[:2])), ("split '--'", lambda: key.split(s, "--")), ("rsplitn 3 ','", lambda: key.rsplitn(s, 3, ",")),
                 ("split_ascii_whitespace", lambda: key.split_ascii_whitespace(s)),
                 ("replace ', ' -> '; '", lambda: key.replace(s, ", ", "; ")),
                 ("replace ',' -> ' and '", lambda: key.replace(s, ",", " and "))]:
    fn()
    key.stats(reset=True)
    print(f"--- {name}", file=sys.stderr)
    t0 = time.perf_counter()
    r = fn()
    dt = time.perf_counter() - t0
    del r
    st = key.stats()
    print(f"{name}: {dt * 1e3:.1f} ms, {st['pbs']} PBS, {st['launches']} launches", file=sys.stderr)
