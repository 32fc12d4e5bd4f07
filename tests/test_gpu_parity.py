"""GPU parity tests proper (-m gpu): the HIP path, called through the C-ABI, against the CPU
oracle on the same seeded inputs -- bit-exact (integer work and the fixed-schedule f64 FFT), plus
decrypt-level checks mirroring the reference's tests
(tfhe/src/shortint/server_key/tests/shortint.rs:366-461 `shortint_keyswitch_programmable_bootstrap`,
core_crypto/algorithms/test/lwe_programmable_bootstrapping.rs:70-166)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def T():
    import tfhe_rs_string_amd as T

    return T


@pytest.fixture(scope="module")
def ctx(T):
    return T.Context(0)


def make_sk(T, ctx, ks):
    p = T.PARAM_MESSAGE_2_CARRY_2_KS_PBS(lwe_dim=ks.p.lwe_dim)
    return T.ServerKey(ctx, p, ks.bsk, ks.ksk)


@pytest.fixture(scope="module", params=[5, 4, 3, 2, 1, 0],
                ids=["quad_kernel", "half_kernel", "block_kernel", "pair_kernel", "wave_kernel", "lds_kernel"])
def small(request, T, ctx, keyset_small):
    """All blind-rotation kernels (5 = four wavefronts per ciphertext, 8 points per thread, one LDS transposition per transform:
    built and measured in round 3, selectable, not on the default path; 4 = 256-thread workgroup per ciphertext, two per CU, the default for 257-512 ciphertexts;
    3 = 512-thread workgroup per ciphertext, the default for small batches; 2 = wave pair per
    ciphertext, the default for large ones; 1 = wave per ciphertext; 0 = simple LDS kernel) go through every parity case."""
    sk = make_sk(T, ctx, keyset_small)
    sk.set_variant(request.param)
    return keyset_small, sk


LUT_FUNCS = [
    lambda x: x % 4,                 # message extract (server_key/mod.rs:619)
    lambda x: x // 4,                # carry extract (server_key/mod.rs:539)
    lambda x: int((x // 4) == (x % 4)),  # bivariate x == y packed as x*4+y (bivariate_pbs.rs:71-131)
    lambda x: (x * x) % 16,
]


def luts_for(T, p):
    return np.stack([T.generate_lookup_table(p, f)[0] for f in LUT_FUNCS])


def test_bsk_to_fourier_bitexact(small):
    ks, sk = small
    re, im = sk.fourier_bsk()
    assert np.array_equal(re, ks.bsk_re) and np.array_equal(im, ks.bsk_im)


@pytest.mark.parametrize("B", [1, 7, 8, 37, 300])
def test_keyswitch_bitexact(small, B):
    ks, sk = small
    cts = ks.encrypt(np.arange(B) % 16)
    got = sk.keyswitch_batch(cts)
    want = ks.keyswitch_batch(cts, threads=4)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("B", [1, 5, 64])
def test_pbs_bitexact(T, small, B):
    ks, sk = small
    cts = ks.encrypt(np.arange(B) % 16)
    small_cts = ks.keyswitch_batch(cts)
    luts = luts_for(T, sk.params)
    idx = (np.arange(B) % len(LUT_FUNCS)).astype(np.uint32)
    got = sk.pbs_batch(small_cts, luts, idx)
    want = ks.pbs_batch(small_cts, luts, idx, threads=4)
    assert np.array_equal(got, want)


def test_ks_pbs_bitexact_and_decrypts(T, small):
    ks, sk = small
    B = 96
    msgs = np.arange(B) % 16
    cts = ks.encrypt(msgs)
    luts = luts_for(T, sk.params)
    idx = (np.arange(B) % len(LUT_FUNCS)).astype(np.uint32)
    got = sk.ks_pbs_batch(cts, luts, idx)
    want = ks.ks_pbs_batch(cts, luts, idx, threads=4)
    assert np.array_equal(got, want)
    dec = ks.decrypt(got)
    assert [int(d) for d in dec] == [LUT_FUNCS[i % 4](int(m)) for i, m in enumerate(msgs)]


@pytest.mark.parametrize("B", [255, 256, 257, 511, 512, 513, 767, 768, 769, 1027, 1400, 1700, 1793, 2049])
def test_automatic_kernel_choice_bitexact(T, ctx, keyset_small, B):
    """default variant (-1): one ciphertext per CU -> variant 3, two -> variant 5 (odd counts too), three -> 5 then 3, beyond
    whole rounds of the wave-pair kernel plus a remainder by the same rule (1400 = 1024 + 376, 1700 = 1024 + 512 + 164);
    ragged batch sizes around the workgroup (4 ciphertexts) and the switch-over boundaries"""
    ks = keyset_small
    sk = make_sk(T, ctx, ks)
    rng = np.random.default_rng(B)
    cts = ks.encrypt(rng.integers(0, 16, B))
    luts = luts_for(T, sk.params)
    idx = rng.integers(0, len(LUT_FUNCS), B).astype(np.uint32)
    got = sk.ks_pbs_batch(cts, luts, idx)
    assert np.array_equal(got, ks.ks_pbs_batch(cts, luts, idx, threads=8))


def test_large_ragged_batch_bitexact(T, ctx, keyset_small):
    """a batch well beyond one round of the chip (5 003 ciphertexts: 20 rounds of the throughput kernel, one keyswitch
    split) stays bit-exact; the level sizes of `contains` on long strings are of this order"""
    ks = keyset_small
    sk = make_sk(T, ctx, ks)
    rng = np.random.default_rng(5003)
    B = 5003
    cts = rng.integers(0, 2**64, (B, sk.params.big_size), dtype=np.uint64)
    luts = luts_for(T, sk.params)
    idx = rng.integers(0, len(LUT_FUNCS), B).astype(np.uint32)
    got = sk.ks_pbs_batch(cts, luts, idx)
    assert np.array_equal(got, ks.ks_pbs_batch(cts, luts, idx, threads=16))


def test_pinned_buffers_and_pipelined_host_batches(T, ctx, keyset_small):
    """buffers from tfhe_hip_host_alloc: no staging copy, and a batch of >= 2560 ciphertexts is pipelined in chunks
    on two streams (upload / kernels / download overlap); same bits as the pageable path and the oracle"""
    ks = keyset_small
    sk = make_sk(T, ctx, ks)
    rng = np.random.default_rng(4099)
    B = 4099  # three chunks of 1024 and a last one of 1027
    pin_in, pin_out = T.PinnedArray((B, sk.params.big_size)), T.PinnedArray((B, sk.params.big_size))
    luts = luts_for(T, sk.params)
    pin_luts = T.PinnedArray(luts.shape)
    pin_luts.array[:] = luts
    pin_in.array[:] = rng.integers(0, 2**64, (B, sk.params.big_size), dtype=np.uint64)
    idx = rng.integers(0, len(LUT_FUNCS), B).astype(np.uint32)
    got = sk.ks_pbs_batch(pin_in.array, pin_luts.array, idx, out=pin_out.array)
    want = ks.ks_pbs_batch(pin_in.array, luts, idx, threads=16)
    assert np.array_equal(got, want)
    assert np.array_equal(sk.ks_pbs_batch(np.array(pin_in.array), luts, idx), want)  # pageable buffers
    small_out = sk.ks_pbs_batch(pin_in.array[:33], pin_luts.array, idx[:33], out=pin_out.array[:33])  # pinned, not pipelined
    assert np.array_equal(small_out, want[:33])
    for b in (2559, 2560, 3071):  # the last size that is not pipelined; one chunk of 1024 and a last one of 1536 / 2047
        pin_out.array[:b] = 0
        assert np.array_equal(sk.ks_pbs_batch(pin_in.array[:b], pin_luts.array, idx[:b], out=pin_out.array[:b]), want[:b]), b


def test_edge_cases_bitexact(T, small):
    """zero mask elements (bootstrap.rs:281 skip), modulus switch hitting 2N (common.rs:20-25),
    all-zero mask (trivial-looking ciphertext), body extremes."""
    ks, sk = small
    p = sk.params
    n = p.lwe_dim
    rng = np.random.default_rng(9)
    cts = rng.integers(0, 2**64, (6, n + 1), dtype=np.uint64)
    cts[0, : n // 2] = 0
    cts[1, :] = np.uint64(2**64 - 1)          # every a_hat == 2N
    cts[2, :n] = 0                            # no rotation except the body
    cts[3, n] = 0
    cts[4, :] = np.uint64(1) << np.uint64(63)  # a_hat == N: pure negation
    cts[5, :] = (np.uint64(1) << np.uint64(52)) - np.uint64(1)  # rounds to 0
    luts = luts_for(T, p)
    idx = np.array([0, 1, 2, 3, 0, 1], dtype=np.uint32)
    got = sk.pbs_batch(cts, luts, idx)
    want = ks.pbs_batch(cts, luts, idx)
    assert np.array_equal(got, want)


def test_device_buffer_path_matches_host_path(T, small):
    import torch

    ks, sk = small
    B = 40
    cts = ks.encrypt(np.arange(B) % 16)
    luts = luts_for(T, sk.params)
    idx = (np.arange(B) % 4).astype(np.uint32)
    want = sk.ks_pbs_batch(cts, luts, idx)
    dev = torch.device("cuda:0")
    d_in = torch.from_numpy(cts.view(np.int64)).to(dev)
    d_luts = torch.from_numpy(luts.view(np.int64)).to(dev)
    d_idx = torch.from_numpy(idx.view(np.int32)).to(dev)
    d_out = torch.empty_like(d_in)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        sk.ks_pbs_batch_device(d_in, d_luts, d_idx, d_out)
    side.synchronize()
    assert np.array_equal(d_out.cpu().numpy().view(np.uint64), want)


def test_errors(T, ctx, small):
    ks, sk = small
    p = sk.params
    with pytest.raises(T.TfheHipError):
        T.ServerKey(ctx, p, ks.bsk[:-1], ks.ksk)      # LweBootstrapKey::from_container length assert
    with pytest.raises(T.TfheHipError):
        T.ServerKey(ctx, p, ks.bsk, ks.ksk[:-8])
    cts = ks.encrypt([1, 2])
    luts = luts_for(T, p)
    with pytest.raises(T.TfheHipError):
        sk.ks_pbs_batch(cts, luts, np.array([0, 99], dtype=np.uint32))  # lut index out of range
    with pytest.raises(ValueError):
        sk.ks_pbs_batch(cts[:, :-1], luts, 0)
    assert sk.ks_pbs_batch(np.zeros((0, p.big_size), dtype=np.uint64), luts, 0).shape == (0, p.big_size)


@pytest.fixture(scope="module")
def full(T, ctx, keyset_full):
    return keyset_full, make_sk(T, ctx, keyset_full)


def test_full_params_bitexact_sample(T, full):
    """PARAM_MESSAGE_2_CARRY_2_KS_PBS, n = 742: a few ciphertexts bit-compared with the oracle."""
    ks, sk = full
    B = 6
    msgs = np.array([0, 3, 7, 9, 12, 15])
    cts = ks.encrypt(msgs)
    luts = luts_for(T, sk.params)
    idx = (np.arange(B) % 4).astype(np.uint32)
    got = sk.ks_pbs_batch(cts, luts, idx)
    want = ks.ks_pbs_batch(cts, luts, idx, threads=6)
    assert np.array_equal(got, want)
    # the automatic choice took the latency kernel (B <= 3 x CUs); the throughput kernels give the same bits
    for v in (5, 4, 3, 2, 1):
        sk.set_variant(v)
        assert np.array_equal(sk.ks_pbs_batch(cts, luts, idx), want), v
    sk.set_variant(-1)


def test_full_params_random_inputs_bitexact_all_kernels(T, full):
    """n = 742, arbitrary (uniformly random) LWE inputs and LUT polynomials -- not just valid encryptions --
    through the keyswitch and every blind-rotation kernel: bit-identical to the oracle"""
    ks, sk = full
    rng = np.random.default_rng(77)
    B = 40
    cts = rng.integers(0, 2**64, (B, sk.params.big_size), dtype=np.uint64)
    luts = rng.integers(0, 2**64, (3, sk.params.lut_len), dtype=np.uint64)
    idx = rng.integers(0, 3, B).astype(np.uint32)
    want_small = ks.keyswitch_batch(cts, threads=8)
    assert np.array_equal(sk.keyswitch_batch(cts), want_small)
    want = ks.pbs_batch(want_small, luts, idx, threads=8)
    for v in (5, 4, 3, 2, 1, 0):
        sk.set_variant(v)
        assert np.array_equal(sk.pbs_batch(want_small, luts, idx), want), v
    sk.set_variant(-1)
    assert np.array_equal(sk.ks_pbs_batch(cts, luts, idx), want)


@pytest.mark.parametrize("B", [1000, 1024, 1027, 2049])
def test_full_size_keyswitch_grid_bitexact(T, ctx, full, monkeypatch, B):
    """n = 742 at the batch sizes the bench and the big op-graph levels run: the matrix-core keyswitch's full grid
    (24 column tiles x ciphertext groups x row splits with the XCD-aware mapping, keyswitch_mfma.hip) on uniformly random
    full-range inputs == the dot4 kernels == the oracle's keyswitch (lwe_keyswitch.rs:96-170), bit for bit"""
    ks, sk = full
    monkeypatch.setenv("TFHE_HIP_KS_MFMA_MIN", "0")
    sk_dot4 = make_sk(T, ctx, ks)
    monkeypatch.delenv("TFHE_HIP_KS_MFMA_MIN")
    rng = np.random.default_rng(B)
    cts = rng.integers(0, 2**64, (B, sk.params.big_size), dtype=np.uint64)
    cts[0, :8] = [0, 1 << 63, 2**64 - 1, 1 << 48, (1 << 48) - 1, (1 << 49) - 1, 1 << 49, 3 << 47]
    cts[B - 1] = 2**64 - 1
    cts[B // 2] = 0
    got = sk.keyswitch_batch(cts)
    assert np.array_equal(got, sk_dot4.keyswitch_batch(cts))
    assert np.array_equal(got, ks.keyswitch_batch(cts, threads=16))


def test_two_ciphertexts_per_cu_soak_all_kernels_agree(T, full):
    """n = 742, levels of 257 ... 512 bootstraps (two workgroups of the quad kernel co-resident on every CU: the automatic
    choice) on uniformly random inputs, 24 launches: the quad kernel == the wave-pair kernel == the half kernel, bit for bit
    -- barrier / LDS-reuse hazards of the time-shared images would show up as run-to-run differences"""
    ks, sk = full
    import torch

    cus = torch.cuda.get_device_properties(0).multi_processor_count
    rng = np.random.default_rng(512)
    luts = rng.integers(0, 2**64, (3, sk.params.lut_len), dtype=np.uint64)
    try:
        for it in range(24):
            B = int(rng.integers(cus + 1, 2 * cus + 1))
            small = rng.integers(0, 2**64, (B, sk.params.small_size), dtype=np.uint64)
            if it % 6 == 0:
                small[:, : sk.params.lwe_dim // 2] = 0  # long runs of rotations by zero
            idx = rng.integers(0, 3, B).astype(np.uint32)
            sk.set_variant(2)
            want = sk.pbs_batch(small, luts, idx)
            sk.set_variant(5)
            got = sk.pbs_batch(small, luts, idx)
            assert np.array_equal(got, want), (it, B)
            assert np.array_equal(sk.pbs_batch(small, luts, idx), want), (it, B, "second run")
            if it % 8 == 0:
                sk.set_variant(4)
                assert np.array_equal(sk.pbs_batch(small, luts, idx), want), (it, B, "half kernel")
            sk.set_variant(-1)
            assert np.array_equal(sk.pbs_batch(small, luts, idx), want), (it, B, "automatic")
    finally:
        sk.set_variant(-1)
    assert sk.device_status() == 0


def test_full_params_batch_1024_decrypts(T, full):
    """BASELINE config 2: 1024 independent KS-PBS; checked through the size-independent property
    decrypt(PBS_f(enc(m))) == f(m) for every ciphertext, and a second PBS on the outputs
    (noise stays bootstrappable), mirroring shortint.rs:402-430."""
    ks, sk = full
    B = 1024
    msgs = np.arange(B) % 16
    cts = ks.encrypt(msgs)
    luts = luts_for(T, sk.params)
    idx = (np.arange(B) % 4).astype(np.uint32)
    out = sk.ks_pbs_batch(cts, luts, idx)
    dec = ks.decrypt(out)
    want = np.array([LUT_FUNCS[i % 4](int(m)) for i, m in enumerate(msgs)])
    assert np.array_equal(dec.astype(np.int64), want)
    out2 = sk.ks_pbs_batch(out, luts, 3)
    assert np.array_equal(ks.decrypt(out2).astype(np.int64), (want * want) % 16)
    # determinism: same launch twice gives the same bits (tests_cases_unsigned.rs:2109-2114)
    assert np.array_equal(sk.ks_pbs_batch(cts, luts, idx), out)


def test_pbs_ks_order_bitexact(T, ctx, oracle):
    """BootstrapKeyswitch order (PARAM_MESSAGE_2_CARRY_2_PBS_KS: ks 4 levels of base 2^4, ciphertexts
    under the small key; programmable_bootstrap_keyswitch_assign, shortint/server_key/mod.rs:859-932)."""
    ks = oracle.KeySet(oracle.param_message_2_carry_2_pbs_ks(lwe_dim=10), seed=0xB5)
    p = T.PARAM_MESSAGE_2_CARRY_2_PBS_KS(lwe_dim=10)
    sk = T.ServerKey(ctx, p, ks.bsk, ks.ksk)
    B = 20
    msgs = np.arange(B) % 16
    cts = ks.encrypt_small(msgs)
    luts = luts_for(T, p)
    idx = (np.arange(B) % 4).astype(np.uint32)
    got = sk.pbs_ks_batch(cts, luts, idx)
    assert np.array_equal(got, ks.pbs_ks_batch(cts, luts, idx, threads=4))
    assert np.array_equal(sk.apply_lookup_table_batch(cts, luts, idx), got)  # dispatch on pbs_order
    assert [int(d) for d in ks.decrypt_small(got)] == [LUT_FUNCS[i % 4](int(m)) for i, m in enumerate(msgs)]
    # the keyswitch with the (4, 4) decomposition alone
    big = ks.pbs_batch(cts, luts, idx)
    assert np.array_equal(sk.keyswitch_batch(big), ks.keyswitch_batch(big))


def test_concurrent_host_threads_share_one_key(T, ctx, keyset_small):
    """The reference shares `ServerKey` immutably across rayon workers (shortint/engine/mod.rs:23-25);
    here: four host threads, each on its own stream, bootstrap through the same key handle at once."""
    import threading

    import torch

    ks = keyset_small
    sk = make_sk(T, ctx, ks)
    luts = luts_for(T, sk.params)
    dev = torch.device("cuda:0")
    d_luts = torch.from_numpy(luts.view(np.int64)).to(dev)
    jobs = []
    for t in range(4):
        B = 24 + 8 * t
        cts = ks.encrypt((np.arange(B) + t) % 16)
        idx = ((np.arange(B) + t) % 4).astype(np.uint32)
        jobs.append((cts, idx, ks.ks_pbs_batch(cts, luts, idx, threads=4)))
    results = [None] * 4
    errors = []

    def work(t):
        try:
            cts, idx, _ = jobs[t]
            stream = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(stream):
                d_in = torch.from_numpy(cts.view(np.int64)).to(dev, non_blocking=False)
                d_idx = torch.from_numpy(idx.view(np.int32)).to(dev)
                d_out = torch.empty_like(d_in)
                for _ in range(3):
                    sk.ks_pbs_batch_device(d_in, d_luts, d_idx, d_out, stream=stream.cuda_stream)
                stream.synchronize()
                results[t] = d_out.cpu().numpy().view(np.uint64)
        except Exception as e:  # pragma: no cover
            errors.append(e)

    threads = [threading.Thread(target=work, args=(t,)) for t in range(4)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors
    for t in range(4):
        assert np.array_equal(results[t], jobs[t][2])


@pytest.mark.parametrize("g", [2, 3])
def test_multi_bit_pbs_bitexact_and_decrypts(T, ctx, oracle, g):
    """multi-bit PBS (grouping factor 2 / 3: PARAM_MULTI_BIT_MESSAGE_2_CARRY_2_GROUP_{2,3}_KS_PBS with a reduced
    lwe_dim): GPU == oracle bit for bit, and decrypt(PBS_f(enc(m))) == f(m)
    (lwe_multi_bit_programmable_bootstrapping.rs doc-test :196-293 checks exactly that property)"""
    OP = oracle.param_multi_bit_message_2_carry_2_group_2 if g == 2 else oracle.param_multi_bit_message_2_carry_2_group_3
    TP = T.PARAM_MULTI_BIT_MESSAGE_2_CARRY_2_GROUP_2_KS_PBS if g == 2 else T.PARAM_MULTI_BIT_MESSAGE_2_CARRY_2_GROUP_3_KS_PBS
    n = 12
    ks = oracle.MultiBitKeySet(OP(lwe_dim=n), g, seed=21)
    p = TP(lwe_dim=n)
    sk = T.ServerKey(ctx, p, ks.bsk, ks.ksk, grouping_factor=g)
    B = 37
    msgs = np.arange(B) % 16
    cts = ks.encrypt(msgs)
    luts = luts_for(T, p)
    idx = (np.arange(B) % len(LUT_FUNCS)).astype(np.uint32)
    got = sk.ks_pbs_batch(cts, luts, idx)
    assert np.array_equal(got, ks.ks_pbs_batch(cts, luts, idx, threads=8))
    assert [int(d) for d in ks.decrypt(got)] == [LUT_FUNCS[i % 4](int(m)) for i, m in enumerate(msgs)]
    small = ks.keyswitch_batch(cts)
    assert np.array_equal(sk.keyswitch_batch(cts), small)
    assert np.array_equal(sk.pbs_batch(small, luts, idx), ks.pbs_batch(small, luts, idx, threads=8))


def test_multi_bit_full_parameters_decrypt(T, ctx, oracle):
    """PARAM_MULTI_BIT_MESSAGE_2_CARRY_2_GROUP_2_KS_PBS at full size (n = 818): a sample bit-compared with the
    oracle, 256 ciphertexts decrypt-checked, and the result bootstraps again"""
    g = 2
    ks = oracle.MultiBitKeySet(oracle.param_multi_bit_message_2_carry_2_group_2(), g, seed=22)
    p = T.PARAM_MULTI_BIT_MESSAGE_2_CARRY_2_GROUP_2_KS_PBS()
    sk = T.ServerKey(ctx, p, ks.bsk, ks.ksk, grouping_factor=g)
    B = 256
    msgs = np.arange(B) % 16
    cts = ks.encrypt(msgs)
    luts = luts_for(T, p)
    idx = (np.arange(B) % 4).astype(np.uint32)
    out = sk.ks_pbs_batch(cts, luts, idx)
    want = np.array([LUT_FUNCS[i % 4](int(m)) for i, m in enumerate(msgs)])
    assert np.array_equal(ks.decrypt(out).astype(np.int64), want)
    assert np.array_equal(out[:4], ks.ks_pbs_batch(cts[:4], luts, idx[:4], threads=4))
    out2 = sk.ks_pbs_batch(out, luts, 3)
    assert np.array_equal(ks.decrypt(out2).astype(np.int64), (want * want) % 16)


def test_host_entry_points_from_concurrent_threads(T, small):
    """the boundary must be re-entrant from many host threads sharing one server key (the reference shares its
    ServerKey between rayon workers): same bits as the oracle from every thread"""
    import threading

    ks, sk = small
    p = sk.params
    luts = luts_for(T, p)
    rng = np.random.default_rng(77)
    jobs = []
    for t in range(6):
        B = int(rng.integers(1, 300))
        cts = rng.integers(0, 2**64, (B, p.big_size), dtype=np.uint64)
        idx = rng.integers(0, len(LUT_FUNCS), B).astype(np.uint32)
        jobs.append((cts, idx, ks.ks_pbs_batch(cts, luts, idx, threads=16)))
    bad = []

    def work(cts, idx, want):
        for _ in range(4):
            if not np.array_equal(sk.ks_pbs_batch(cts, luts, idx), want):
                bad.append(cts.shape[0])

    threads = [threading.Thread(target=work, args=j) for j in jobs]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not bad, bad


@pytest.mark.parametrize("B", [1, 31, 32, 33, 129, 1000])
def test_the_two_keyswitch_kernels_agree_with_each_other_and_the_oracle(T, ctx, keyset_small, monkeypatch, B):
    """the matrix-core keyswitch (keyswitch_mfma.hip, the default) and the dot4 kernels (kernels.hip) on the same batches:
    tile edges (31 / 32 / 33 ciphertexts), random full-range inputs and the extreme coefficients (0, 2^63, 2^64 - 1: the
    rounding of the decomposition), against each other and the oracle's keyswitch (lwe_keyswitch.rs:96-170)"""
    ks = keyset_small
    sk = make_sk(T, ctx, ks)
    monkeypatch.setenv("TFHE_HIP_KS_MFMA_MIN", "0")
    sk_dot4 = make_sk(T, ctx, ks)
    monkeypatch.delenv("TFHE_HIP_KS_MFMA_MIN")
    rng = np.random.default_rng(B)
    cts = rng.integers(0, 2**64, (B, sk.params.big_size), dtype=np.uint64)
    cts[0, :8] = [0, 1 << 63, 2**64 - 1, 1 << 48, (1 << 48) - 1, (1 << 49) - 1, 1 << 49, 3 << 47]
    if B > 2:
        cts[B - 1] = 2**64 - 1
        cts[B // 2] = 0
    got = sk.keyswitch_batch(cts)
    assert np.array_equal(got, sk_dot4.keyswitch_batch(cts))
    assert np.array_equal(got, ks.keyswitch_batch(cts, threads=8))


GENERIC_CASES = [
    # (lwe_dim, poly_size, pbs_base_log, pbs_level): the generic kernel (variant 0) beyond N = 2048 / one level of 2^23
    (1, 256, 23, 1),     # PARAM_MESSAGE_2_CARRY_2_TEST, the reference string layer's own fast-test set (ciphertext.rs:75-90)
    (12, 1024, 23, 1),   # N/2 = 2 * 4^4: one radix-2 stage in front of the radix-4 ones
    (5, 512, 15, 2),     # two PBS levels
    (3, 4096, 23, 1),    # N/2 = 2 * 4^5, 128 KiB of LDS per workgroup
    (4, 2048, 12, 3),    # N = 2048 with a decomposition the wave / workgroup kernels are not built for
]


@pytest.mark.parametrize("n,N,bl,lv", GENERIC_CASES)
def test_generic_parameter_sets_bitexact(T, ctx, oracle, n, N, bl, lv):
    """k = 1, any power-of-two polynomial size 256 .. 4096, any PBS decomposition: keyswitch (matrix cores), key conversion and
    the generic blind rotation against the oracle, bit for bit, plus decrypt(PBS_f(enc(m))) == f(m) where the noise allows"""
    op = oracle.param_custom(n, N, bl, lv)
    ks = oracle.KeySet(op, seed=1000 + N + n)
    p = T.Params(n, 1, N, bl, lv, 3, 5, 4, 4, 0)
    sk = T.ServerKey(ctx, p, ks.bsk, ks.ksk)
    re, im = sk.fourier_bsk()
    assert np.array_equal(re, ks.bsk_re) and np.array_equal(im, ks.bsk_im)
    B = 21
    msgs = np.arange(B) % 16
    cts = ks.encrypt(msgs)
    luts = luts_for(T, p)
    idx = (np.arange(B) % len(LUT_FUNCS)).astype(np.uint32)
    small = ks.keyswitch_batch(cts)
    assert np.array_equal(sk.keyswitch_batch(cts), small)
    got = sk.ks_pbs_batch(cts, luts, idx)
    assert np.array_equal(got, ks.ks_pbs_batch(cts, luts, idx, threads=8))
    rng = np.random.default_rng(N)
    rnd = rng.integers(0, 2**64, (7, p.small_size), dtype=np.uint64)  # arbitrary inputs, zeros included
    rnd[2, : max(1, n // 2)] = 0
    assert np.array_equal(sk.pbs_batch(rnd, luts, idx[:7]), ks.pbs_batch(rnd, luts, idx[:7], threads=4))
    assert [int(d) for d in ks.decrypt(got)] == [LUT_FUNCS[i % 4](int(m)) for i, m in enumerate(msgs)]
    with pytest.raises(T.TfheHipError, match="variants"):  # the wave / workgroup kernels are built for N = 2048, 23 x 1 only
        sk.set_variant(2)


GLWE_DIM_CASES = [
    # (name, lwe_dim, k, N, pbs base log, pbs level, ks base log, ks level, message modulus, carry modulus, lwe std, glwe std):
    # the classic sets of shortint/parameters/mod.rs:599-660 with k > 1 (lwe_dim cut down where the oracle should run in
    # seconds; one set at its full size)
    ("1_0", 10, 5, 256, 15, 1, 5, 2, 2, 1, 0.000022810107419132102, 0.00000000037411618952047216),
    ("1_1", 9, 3, 512, 18, 1, 4, 3, 2, 2, 0.00002043784477291318, 0.0000000000034525330484572114),
    ("2_0", 8, 2, 512, 8, 2, 3, 4, 4, 1, 0.000034119201269311964, 0.00000004053919869756513),
    ("1_2", 7, 2, 1024, 23, 1, 4, 3, 2, 4, 0.000007069849454709433, 0.00000000000000029403601535432533),
    ("1_1_full_size", 684, 3, 512, 18, 1, 4, 3, 2, 2, 0.00002043784477291318, 0.0000000000034525330484572114),
]


@pytest.mark.parametrize("name,n,k,N,bl,lv,kbl,klv,mm,cm,lstd,gstd", GLWE_DIM_CASES, ids=[c[0] for c in GLWE_DIM_CASES])
def test_glwe_dimension_above_one_bitexact(T, ctx, oracle, name, n, k, N, bl, lv, kbl, klv, mm, cm, lstd, gstd):
    """k = 2, 3, 5 (PARAM_MESSAGE_1_CARRY_0 / 1_1 / 2_0 / 1_2 _KS_PBS): keyswitch with their decompositions, key conversion,
    the generic blind rotation and the sample extraction over k mask polynomials against the oracle, bit for bit, plus
    decrypt(PBS_f(enc(m))) == f(m)"""
    op = oracle.param_custom(n, N, bl, lv, kbl, klv, glwe_dim=k, message_modulus=mm, carry_modulus=cm, lwe_std=lstd, glwe_std=gstd)
    ks = oracle.KeySet(op, seed=2000 + N + k)
    p = T.Params(n, k, N, bl, lv, kbl, klv, mm, cm, 0)
    assert p.big_size == k * N + 1 and p.lut_len == (k + 1) * N
    sk = T.ServerKey(ctx, p, ks.bsk, ks.ksk)
    re, im = sk.fourier_bsk()
    assert np.array_equal(re, ks.bsk_re) and np.array_equal(im, ks.bsk_im)
    space = mm * cm
    funcs = [lambda x: x % mm, lambda x: x // mm, lambda x: (x * x) % space, lambda x: (space - 1 - x)]
    luts = np.stack([T.generate_lookup_table(p, f)[0] for f in funcs])
    assert np.array_equal(luts, np.stack([ks.lut([f(x) for x in range(space)])[0] for f in funcs]))
    B = 19
    msgs = np.arange(B) % space
    cts = ks.encrypt(msgs)
    idx = (np.arange(B) % 4).astype(np.uint32)
    small = ks.keyswitch_batch(cts)
    assert np.array_equal(sk.keyswitch_batch(cts), small)
    got = sk.ks_pbs_batch(cts, luts, idx)
    assert np.array_equal(got, ks.ks_pbs_batch(cts, luts, idx, threads=8))
    rng = np.random.default_rng(N + k)
    rnd = rng.integers(0, 2**64, (7, p.small_size), dtype=np.uint64)
    rnd[3, : max(1, n // 2)] = 0
    assert np.array_equal(sk.pbs_batch(rnd, luts, idx[:7]), ks.pbs_batch(rnd, luts, idx[:7], threads=4))
    assert [int(d) for d in ks.decrypt(got)] == [funcs[i % 4](int(m)) for i, m in enumerate(msgs)]
    with pytest.raises(T.TfheHipError, match="variants"):
        sk.set_variant(3)


def test_every_classic_parameter_set_of_the_table_loads_and_bootstraps(T, ctx):
    """the 15 classic KS_PBS sets of T.CLASSIC_KS_PBS at their FULL sizes (synthetic keys: this is the parameter check, the key
    conversion, the keyswitch decompositions and the kernels' launch shapes, not a parity test -- parity is the cases above),
    and the refusal of a 6-bit set"""
    rng = np.random.default_rng(15)
    for name in T.CLASSIC_KS_PBS:
        p = T.classic_params(name)
        sk = T.ServerKey(ctx, p, rng.integers(0, 2**64, p.bsk_len, dtype=np.uint64), rng.integers(0, 2**64, p.ksk_len, dtype=np.uint64))
        lut, _ = T.generate_lookup_table(p, lambda x: x % p.message_modulus)
        cts = rng.integers(0, 2**64, (3, p.big_size), dtype=np.uint64)
        out = sk.ks_pbs_batch(cts, lut, 0)
        assert out.shape == (3, p.big_size) and sk.device_status() == 0, name
        del sk
    with pytest.raises(T.TfheHipError, match="poly_size"):
        p = T.Params(915, 1, 8192, 22, 1, 3, 5, 64, 1, 0)
        T.ServerKey(ctx, p, np.zeros(p.bsk_len, dtype=np.uint64), np.zeros(p.ksk_len, dtype=np.uint64))


def test_reference_test_parameter_set_loads(T, ctx, oracle):
    """PARAM_MESSAGE_2_CARRY_2_TEST by name: keys from the oracle, a string-sized batch through the default dispatch"""
    ks = oracle.KeySet(oracle.param_message_2_carry_2_test(), seed=77)
    p = T.PARAM_MESSAGE_2_CARRY_2_TEST()
    sk = T.ServerKey(ctx, p, ks.bsk, ks.ksk)
    B = 300
    msgs = np.arange(B) % 16
    cts = ks.encrypt(msgs)
    luts = luts_for(T, p)
    idx = (np.arange(B) % 4).astype(np.uint32)
    got = sk.ks_pbs_batch(cts, luts, idx)
    assert np.array_equal(got, ks.ks_pbs_batch(cts, luts, idx, threads=8))
    assert [int(d) for d in ks.decrypt(got)] == [LUT_FUNCS[i % 4](int(m)) for i, m in enumerate(msgs)]
