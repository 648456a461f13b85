"""Worker for tests/test_driver_distributed.py: one rank of a gloo job running the render driver
with an oracle-backed stand-in renderer (CPU).  argv: scene.xml out.png result.npy"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from raytracer_amd import capi, driver  # noqa: E402
import oracle_lib  # noqa: E402


class OracleRenderer:
    """Same interface as driver.HipRenderer, computing with the CPU oracle (test stand-in only)."""
    device = torch.device("cpu")

    def photon_pass(self, scene, part, n_parts):
        self._photons = oracle_lib.Photons(scene, part, n_parts)
        self._atlas = torch.from_numpy(self._photons.atlas())
        return self._atlas.reshape(-1)       # a view: the all-reduce lands in self._atlas

    def render(self, scene, bands=None, photons_ready=False):
        if photons_ready:
            self._photons.set_atlas(self._atlas.numpy())
            photons = self._photons
        else:
            photons = oracle_lib.Photons(scene) if scene.desc.photon_steps > 0 else None
        h, w = scene.height, scene.width
        if bands is None:
            img, _ = oracle_lib.render(scene, photons, threads=2)
            return torch.from_numpy(img.reshape(-1, 4).copy())
        rows = driver.band_rows_of(h, bands.first_band, bands.band_stride, bands.band_rows)
        out = np.zeros((len(rows), w, 4), dtype=np.float32)
        for i, y in enumerate(rows):
            img, _ = oracle_lib.render(scene, photons, threads=1, rows=(y, y + 1, 1))
            out[i] = img[y]
        return torch.from_numpy(out.reshape(-1, 4))

    def mul_add(self, acc, src, factor):
        acc += src * np.float32(factor)      # two roundings, like Picture::mulAdd

    def quantize(self, rgba):
        return torch.from_numpy(capi.quantize(rgba.numpy()))


def main():
    scene, out, result = sys.argv[1:4]
    dist.init_process_group("gloo")
    frames = driver.render_scene_file(OracleRenderer(), scene, out)
    if dist.get_rank() == 0:
        np.save(result, np.stack(frames))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
