"""GPU, >= 2 devices (self-skipping on the 1-GPU test box): the two multi-GPU arrangements on real hardware.

(1) one process per GPU (the bench / torchrun arrangement): every rank encrypts its block of a 3072-bit (cfg4) and a
    4096-bit (cfg5) batch, ciphertext bits are checked against the oracle with explicit randomness, shards are
    all-gathered over RCCL (sharding.gather_rows), and every rank decrypts the gathered batch;
(2) one process, one key replicated over the devices (engine.fan_out / pai_scatter / pai_gather): same bits as the
    single-device path."""
import json
import os
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import paillier_oracle as orc

pytestmark = pytest.mark.gpu

NDEV = torch.cuda.device_count() if torch.cuda.is_available() else 0
needs2 = pytest.mark.skipif(NDEV < 2, reason="needs at least two GPUs")


def _key(bits):
    fx = json.loads((Path(__file__).parent / "golden" / "fixture_keys.json").read_text())[str(bits)]
    return orc.make_key(int(fx["p"], 16), int(fx["q"], 16), djn_x=(1 << 70) + 12345, bits=bits)


def _rank_main(rank, world, port, bits, n_total, q):
    import torch.distributed as dist

    from pailliercryptolib_python_amd import engine, sharding

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    key = _key(bits)
    pub = engine.PublicKeyHandle(key.n, bits, key.hs, key.randbits, device=dev)
    priv = engine.PrivateKeyHandle(pub, key.p, key.q)
    rng = np.random.default_rng(bits)
    m_all = [int.from_bytes(rng.bytes(bits // 8 + 8), "little") % key.n for _ in range(n_total)]
    r_all = orc.synth_r_limbs(4000 + bits, n_total, key.randbits)
    b, c = engine.shard_plan(n_total, world)[rank]
    assert (b, b + c) == sharding.my_shard(n_total, rank, world)
    m_loc = engine.to_device_words(engine.ints_to_words(m_all[b:b + c], pub.n_words), dev)
    r_loc = engine.to_device_words(r_all[b:b + c], dev)
    ct_loc = pub.encrypt(m_loc, r_loc)
    ok = True
    for i in sorted({0, c // 2, c - 1}):                       # ciphertext bits of this rank's block vs the oracle
        want = orc.encrypt(key, m_all[b + i], orc.limbs_to_ints(r_all[b + i:b + i + 1])[0])
        ok = ok and engine.words_to_ints(engine.to_host_words(ct_loc[i:i + 1]))[0] == want
    full = sharding.gather_rows(ct_loc, n_total)               # RCCL all-gather over xGMI
    ok = ok and bool(torch.equal(full[b:b + c], ct_loc))
    back = priv.decrypt(full.contiguous())
    ok = ok and engine.words_to_ints(engine.to_host_words(back)) == m_all
    q.put((rank, ok))
    dist.destroy_process_group()


@needs2
@pytest.mark.parametrize("bits,n_total", [(3072, 2 * 1024 + 5), (4096, 2 * 512 + 3)])
def test_block_shards_encrypt_gather_decrypt_over_rccl(bits, n_total):
    import torch.multiprocessing as mp

    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + (os.getpid() + bits) % 300
    procs = [ctx.Process(target=_rank_main, args=(r, world, port, bits, n_total, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=120)
    assert sorted(res) == [(r, True) for r in range(world)]


@needs2
def test_single_process_fanout_over_two_real_devices(monkeypatch):
    from pailliercryptolib_python_amd import PaillierPrivateKey, PaillierPublicKey, bindings
    from pailliercryptolib_python_amd.bindings import ipclPublicKey

    monkeypatch.setattr(bindings, "FANOUT_MIN_PER_DEVICE", 256)
    key = orc.make_key(orc.BENCH_P, orc.BENCH_Q, djn_x=0x1234567, bits=2048)
    N = 256 * NDEV + 77
    x = np.random.default_rng(5).uniform(-1000, 1000, N)
    r = orc.synth_r_limbs(55, N, key.randbits)
    one = PaillierPublicKey(ipclPublicKey(key.n, 2048, True, hs=key.hs, randbits=key.randbits, device="cuda:0"))
    many = PaillierPublicKey(ipclPublicKey(key.n, 2048, True, hs=key.hs, randbits=key.randbits,
                                           devices=[f"cuda:{i}" for i in range(NDEV)]))
    sk = PaillierPrivateKey(many, orc.BENCH_P, orc.BENCH_Q)
    a, b = one.encrypt(x, r=r), many.encrypt(x, r=r)
    assert torch.equal(a.words, b.words) and b.words.device == torch.device("cuda", 0)
    want = orc.api_encrypt(key, list(x[:3]) + [x[-1]], orc.limbs_to_ints(np.concatenate([r[:3], r[-1:]])))[0]
    got = [int(v) for v in b[0:3].ciphertextBN()] + [int(b.ciphertextBN(N - 1))]
    assert got == want
    w = np.random.default_rng(6).uniform(-2, 2, N)
    assert torch.equal((a * w).words, (b * w).words)
    assert np.array_equal(sk.decrypt_to_numpy(b), x)
    assert torch.cuda.current_device() == 0
