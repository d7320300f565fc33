"""Sample-range partition of one frame across ranks (one process per GPU).

Every pixel-sample is independent (the RNG is a pure function of seed, pixel, sample, counter), so rank r of
`world` renders samples [r*spp/world, (r+1)*spp/world) of EVERY pixel into a full-frame float4 accumulator;
one all-reduce(sum) of the accumulators then gives the frame's sums.  The reference's own split is by pixel
ranges over 8 threads (04_PathTracingLights/Main.cpp:399-420); sample ranges balance regardless of the scene."""


def sample_range(rank, world, spp):
    if not (0 <= rank < world) or spp < 0:
        raise ValueError("bad rank/world/spp")
    b, e = rank * spp // world, (rank + 1) * spp // world
    # rt_path_params reads (0, 0) as "the whole frame": an EMPTY share (spp < world) must not look like that, or its
    # rank would render every sample and the all-reduced frame would count them twice.  [spp, spp) is empty and explicit.
    return (b, e) if e > b else (spp, spp)


def all_ranges(world, spp):
    return [sample_range(r, world, spp) for r in range(world)]


def is_empty(rng):
    return rng[1] <= rng[0]
