import sys, time, torch
sys.path.insert(0, '.')
from music2midi_amd import synth
from music2midi_amd.config import T5Geometry, default_config
from oracle.t5 import T5Oracle
cfg = default_config(); g = T5Geometry(cfg.model.t5)
sd = synth.t5_state_dict(g, 0); orc = T5Oracle(g, sd)
x = torch.from_numpy(synth.normal(1, "e", (1, 864, 384), 3.0))
for th in (4, 8, 16, 32, 64, 128):
    torch.set_num_threads(th)
    t0 = time.perf_counter(); enc = orc.encode(x); t1 = time.perf_counter()
    orc.generate(x, 49, enc_out=enc); t2 = time.perf_counter()
    print(th, "threads: encode %.2fs, 48 steps %.2fs -> %.1f tok/s" % (t1 - t0, t2 - t1, 48 / (t2 - t1)), flush=True)
