"""Manual check (CPU, ~2 min; too slow for the test suite): VC.pipeline on 14 s of audio over 2 gloo ranks with RMVPE's U-Net actually
cut over time (1 632 frames, 816 per rank + 320 of context), against the single-process run.  Expected output: both ranks equal,
and equal to the single process (r2: bit-identical int16)."""
import os, sys, numpy as np, torch, torch.distributed as td, torch.multiprocessing as mp
os.environ.setdefault("AICG_DEV", "1")   # development switches are live in tools (aicovergen_amd/_env.py)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
def work(rank, world, port, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port); os.environ["AICG_EMU_THREADS"] = "3"
    torch.set_num_threads(3)
    import conftest; conftest._bind("emu")
    if world > 1: td.init_process_group("gloo", rank=rank, world_size=world)
    import test_pipeline as tp
    from synthetic import weights
    from synthetic.inputs import vocal_like
    from aicovergen_amd import rmvpe
    calls = []
    orig = rmvpe.E2E.features_sharded
    def spy(self, mel, group):
        calls.append((mel.shape[-1], None if group is None else td.get_world_size(group)))
        return orig(self, mel, group)
    rmvpe.E2E.features_sharded = spy
    nets = weights.small_model_set(1234)
    out, _, vc = tp.run(conftest.Dev("emu"), nets, vocal_like(14.0, 16000, 1239), x=(1, 1, 4, 6))
    q.put((rank, out, calls))
    if world > 1: td.destroy_process_group()
if __name__ == "__main__":
    ctx = mp.get_context("spawn"); q = ctx.Queue()
    ps = [ctx.Process(target=work, args=(r, 2, 29733, q)) for r in range(2)]
    for p in ps: p.start()
    ps1 = ctx.Process(target=work, args=(0, 1, 29734, q)); ps1.start()
    res = [q.get(timeout=1500) for _ in range(3)]
    for p in ps + [ps1]: p.join(60)
    outs = {}
    for rank, out, calls in res:
        print("rank", rank, "calls", calls, out.shape)
        outs.setdefault(len(calls) and calls[0][1], []).append(out)
    a = outs[2][0]; b = outs[2][1]; s = outs[None][0] if None in outs else outs[0][0]
    print("ranks equal:", np.array_equal(a, b))
    d = np.abs(a.astype(np.int32) - s.astype(np.int32)); print("vs single: max", d.max(), "<=1 LSB", (d <= 1).mean(), "exact", (d == 0).mean())
