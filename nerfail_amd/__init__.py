"""nerfail_amd: the NeRFail render-and-attack hot path on AMD MI355X (gfx950).

Python mirrors of the reference's call signatures (run_nerf, run_nerf_helpers, nerf_to_coord, GaussNet,
create_index_and_dist, the NeRFail-S step) over the C ABI of libnerfail_hip.so (include/nerfail_hip.h).
There is no CPU implementation in this package: without the HIP library or a GPU, calls raise.
"""
__all__ = ['run_nerf', 'run_nerf_helpers', 'nerf_to_coord', 'GaussNet', 'create_index_and_dist', 'attack',
           'sharding', 'build']
