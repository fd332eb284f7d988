"""N > 1 path on CPU: world_size 2 over gloo.  The distributed plumbing bench.py / the hosts use
(pybsgs.dist: table broadcast, round-robin tile dealing, max / sum reductions) with the oracle's tile model
standing in for the GPU engine.  The union of the ranks' hit lists must equal the single-process result and
the throughput accounting must add up."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import json, os, sys
sys.path.insert(0, os.path.join(ROOT, "bsgs-cuda_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from pybsgs import dist as D
import oracle_lib as O

rank, local_rank, world = D.init("gloo")
fx = json.load(open(os.path.join(ROOT, "tests", "golden", "small_w1024_ht8_t2_b2_p4.json")))
gpu = bytes.fromhex(fx["htgpu"]); g2 = bytes.fromhex(fx["g2"])
# rank 0 owns the table image; everyone else receives it through the one start-up collective
img = torch.from_numpy(np.frombuffer(gpu, dtype=np.int32).copy()) if rank == 0 else torch.zeros(len(gpu) // 4, dtype=torch.int32)
secs = D.broadcast_table(img, src=0)
table = img.numpy().tobytes()
assert table == gpu
tiles = fx["tiles"] + fx["known_key"]["walk"]
mine = D.deal_tiles(list(enumerate(tiles)), rank, world)
hits = []
for k, tl in mine:
    h, n = O.tile_ref((int(tl["px"], 16), int(tl["py"], 16)), g2, fx["t"], fx["b"], fx["p"], table, fx["htsz"])
    assert [list(x) for x in h] == tl["hits"]
    hits += [(k, c, i) for c, i in h]
D.barrier(cuda=False)
# what bench.py --dump-hits gathers: every rank's record, in rank order, on every rank
recs = D.gather_objects({"rank": rank, "tiles": [k for k, _ in mine], "nhits": len(hits)})
assert [r["rank"] for r in recs] == list(range(world)) and recs[rank]["nhits"] == len(hits)
# replica verification (bench.py): equal objects on every rank -> True everywhere; one rank differing in one value -> False everywhere
import hashlib
same, sums = D.all_equal([hashlib.sha256(table).hexdigest(), len(g2)])
bad, _ = D.all_equal([hashlib.sha256(table).hexdigest(), len(g2) + (1 if rank == 1 else 0)])
assert same and not bad and len(sums) == world and D.XGMI_LINK_GBPS > 100
# the "1/N each + all-gather" start-up (bench.py --startup-strategy allgather): every rank holds its own slice of one buffer, in place; afterwards all hold all
full = torch.zeros(world * 4096, dtype=torch.uint8)
full[rank * 4096:(rank + 1) * 4096] = torch.arange(4096, dtype=torch.int32).add(rank * 7).remainder(251).to(torch.uint8)
D.allgather_slices(full, 4096)
for r in range(world):
    assert torch.equal(full[r * 4096:(r + 1) * 4096], torch.arange(4096, dtype=torch.int32).add(r * 7).remainder(251).to(torch.uint8)), r
steps = D.reduce_sum_int(len(mine) * 2 * fx["t"] * fx["b"] * fx["p"])
tmax = D.reduce_max([0.5 + rank])[0]
json.dump({"rank": rank, "world": world, "gathered": recs, "hits": hits, "tiles": [k for k, _ in mine], "steps": steps, "tmax": tmax},
          open(os.path.join(OUT, "r%d.json" % rank), "w"))
import torch.distributed as td
td.destroy_process_group()
'''


def test_two_ranks_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text("ROOT = %r\nOUT = %r\n" % (ROOT, str(tmp_path)) + WORKER)
    import socket
    with socket.socket() as sk:                      # a port nobody holds (a fixed one may sit in TIME_WAIT after a previous run)
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(script)]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)      # three cold `import torch` can take minutes
    assert res.returncode == 0, res.stderr[-2000:]
    r = [json.load(open(tmp_path / ("r%d.json" % k))) for k in range(2)]
    fx = json.load(open(os.path.join(ROOT, "tests", "golden", "small_w1024_ht8_t2_b2_p4.json")))
    tiles = fx["tiles"] + fx["known_key"]["walk"]
    assert sorted(r[0]["tiles"] + r[1]["tiles"]) == list(range(len(tiles)))            # every tile exactly once
    assert r[0]["tiles"] == list(range(0, len(tiles), 2)) and r[1]["tiles"] == list(range(1, len(tiles), 2))
    union = sorted(tuple(h) for h in r[0]["hits"] + r[1]["hits"])
    expect = sorted((k, c, i) for k, tl in enumerate(tiles) for c, i in tl["hits"])
    assert union == expect
    assert r[0]["steps"] == r[1]["steps"] == len(tiles) * 2 * fx["t"] * fx["b"] * fx["p"]
    assert r[0]["tmax"] == r[1]["tmax"] == 1.5                                          # max over ranks
    assert r[0]["gathered"] == r[1]["gathered"] and [g["tiles"] for g in r[0]["gathered"]] == [r[0]["tiles"], r[1]["tiles"]]


def test_bench_gpus_flag_spawns_one_rank_per_gpu():
    """`python bench.py --gpus N` without a launcher re-executes itself under torch.distributed.run with N ranks on 127.0.0.1
    (round 1 parsed --gpus and ignored it); under a launcher (WORLD_SIZE set) it must not spawn again"""
    env = dict(os.environ, BENCH_PRINT_SPAWN="1")
    env.pop("WORLD_SIZE", None)
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "20", "--warmup", "5"], env=env,
                         capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    cmd = json.loads(res.stdout.strip().splitlines()[-1])
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "8" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "8", "--steps", "20", "--warmup", "5"] and cmd[-7].endswith("bench.py")
    # already a rank of a launcher: no second spawn (it goes on to need a GPU)
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8"], env=dict(env, WORLD_SIZE="8", RANK="0", LOCAL_RANK="0"),
                         capture_output=True, text=True, timeout=600)
    assert res.returncode != 0 and "needs an MI355X" in res.stderr


def test_bench_same_device_flag_and_cpu_topology():
    """--same-device (N ranks on cuda:0 over gloo: BASELINE config 5's code path inside a one-GPU lease) spawns N ranks as well and does not
    ask for N GPUs; the CPU baseline reports PHYSICAL cores next to hardware threads"""
    env = dict(os.environ, BENCH_PRINT_SPAWN="1")
    env.pop("WORLD_SIZE", None)
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--same-device", "--w", "26", "--htsz", "25"], env=env,
                         capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    cmd = json.loads(res.stdout.strip().splitlines()[-1])
    assert cmd[cmd.index("--nproc-per-node") + 1] == "2" and "--same-device" in cmd and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    sys.path.insert(0, ROOT)
    import importlib
    bench = importlib.import_module("bench")
    cores, threads = bench.cpu_topology()
    assert 1 <= cores <= threads == (os.cpu_count() or 1)
    pairs = set()
    phys = core = None
    for line in open("/proc/cpuinfo"):
        if line.startswith("physical id"):
            phys = line.split(":")[1].strip()
        elif line.startswith("core id"):
            core = line.split(":")[1].strip()
            pairs.add((phys, core))
    if pairs:
        assert cores == len(pairs)
