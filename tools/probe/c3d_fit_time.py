"""Conv3DModel.fit fed by convolution.DataGenerator(device='cuda') against the bare graphed step."""
import cProfile, pstats, random, time
import torch
from weather_model_amd import synth
from weather_model_amd.dropin import convolution as cv

data = {"T": synth.uniform(12, (40, 1, 24, 50, 60))}
for device in ("cuda", None):
    gen = cv.DataGenerator(data, device=device) if device else cv.DataGenerator(data)
    m = cv.create_model((1, 20, 20, 20))
    cv.compile_model(m)
    random.seed(5)
    m.fit(gen, epochs=1, steps_per_epoch=20, verbose=0)
    torch.cuda.synchronize()
    steps = 300
    t0 = time.perf_counter()
    m.fit(gen, epochs=1, steps_per_epoch=steps, verbose=0)
    torch.cuda.synchronize()
    print(f"generator device={device}: {(time.perf_counter() - t0) / steps * 1e3:.3f} ms per step", flush=True)
    if device:
        pr = cProfile.Profile()
        pr.enable()
        m.fit(gen, epochs=1, steps_per_epoch=steps, verbose=0)
        torch.cuda.synchronize()
        pr.disable()
        pstats.Stats(pr).sort_stats("tottime").print_stats(14)
