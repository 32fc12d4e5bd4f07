#!/usr/bin/env python3
"""(Lives under tests/: it uses the CPU oracle as the client.)  End-to-end example on one MI355X: client keys and encryption by the CPU oracle (test infrastructure standing in
for a tfhe-rs client), server-side string operations through the C-ABI (include/tfhe_hip_strings.h), decryption.
    python tests/manual/demo.py ["some text"] ["pattern"]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import tfhe_rs_string_amd as T  # noqa: E402
from oracle import pyoracle as O  # noqa: E402


def main():
    text = sys.argv[1] if len(sys.argv) > 1 else "  Hello, encrypted World!  "
    pat = sys.argv[2] if len(sys.argv) > 2 else "World"
    print("generating keys (PARAM_MESSAGE_2_CARRY_2_KS_PBS) ...", flush=True)
    ks = O.KeySet(O.param_message_2_carry_2(), seed=2024)
    p = T.PARAM_MESSAGE_2_CARRY_2_KS_PBS()
    sk = T.ServerKey(T.Context(0), p, ks.bsk, ks.ksk)
    key = T.StringServerKey(sk.handle, p)
    enc = lambda s: ks.encrypt([(ord(ch) >> (2 * i)) & 3 for ch in s for i in range(4)]).reshape(len(s), 4, -1)
    dec_bool = lambda b: bool(ks.decrypt(b.reshape(1, -1))[0])
    dec_int = lambda blocks: sum(int(d % 4) << (2 * i) for i, d in enumerate(ks.decrypt(blocks)))

    def dec_str(content):
        d = ks.decrypt(content.reshape(-1, content.shape[-1])).reshape(-1, 4)
        return "".join(chr(int(sum(int(v) << (2 * i) for i, v in enumerate(ch)))) for ch in d).replace("\x00", "")

    es, ep = enc(text), enc(pat)

    def run(name, fn, show):
        key.stats(reset=True)
        t0 = time.perf_counter()
        r = fn()
        dt = (time.perf_counter() - t0) * 1e3
        st = key.stats()
        print(f"{name:34s} -> {show(r)!r:32}  {dt:7.1f} ms, {st['pbs']:5d} PBS in {st['launches']:2d} launches")

    run(f"contains({pat!r})", lambda: key.contains(es, ep), dec_bool)
    run(f"find({pat!r})", lambda: key.find(es, ep), lambda r: (dec_bool(r[0]), dec_int(r[1])))
    run("to_uppercase", lambda: key.to_uppercase(es), dec_str)
    run("trim", lambda: key.trim(es), lambda r: (dec_str(r[0]), dec_int(r[1])))
    run(f"strip_suffix({text[-3:]!r})", lambda: key.strip_suffix(es, text[-3:]), lambda r: (dec_bool(r[0]), dec_str(r[1])))
    run("eq_ignore_case(swapcase)", lambda: key.eq_ignore_case(es, enc(text.swapcase())), dec_bool)
    run("le('zebra')", lambda: key.le(es, "zebra"), dec_bool)

    def dec_split(r):
        parts, number = r
        return [dec_str(x) for x in parts[:dec_int(number)]]

    run("split(' ')", lambda: key.split(es, " "), dec_split)
    run("split_ascii_whitespace", lambda: key.split_ascii_whitespace(es), dec_split)
    run(f"rsplit_once({pat[:1]!r})", lambda: key.rsplit_once(es, pat[:1]), lambda r: (dec_bool(r[0]), dec_str(r[1]), dec_str(r[2])))
    run(f"replace({pat!r}, 'GPU')", lambda: key.replace(es, ep, "GPU"), lambda r: dec_str(r[0]))
    run("substring(enc 2, enc 9)", lambda: key.substring(es, ks.encrypt([2, 0, 0, 0]), ks.encrypt([1, 2, 0, 0])), lambda r: dec_str(r[1]))


if __name__ == "__main__":
    main()
