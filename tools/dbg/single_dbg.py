import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import image_codecs_amd as ica
os.environ["MIJ_GPU_WALK_MIN_PIXELS"] = "0"
for (w, h) in ((512, 512), (1024, 768), (1024, 1024), (1920, 1080)):
    data = ica.stbi_write_jpg_to_memory(ica.synth_rgb(w, h, 1), 90)
    ica.stbi_load_from_memory(data, 3)
    os.environ["MIJ_ES_DEBUG"] = "1"
    os.environ["MIJ_TIMING"] = "1"
    t0 = time.perf_counter()
    ica.stbi_load_from_memory(data, 3)
    print(w, h, "ms", (time.perf_counter() - t0) * 1e3, flush=True)
    os.environ.pop("MIJ_ES_DEBUG")
