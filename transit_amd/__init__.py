"""transit_amd: MI355X-native line-by-line spectrum core behind transit's interfaces.

    host.Problem   options, TLI/atmosphere/CIA readers, samplings  (C++: csrc/host)
    engine.Engine  create/run/destroy over the HIP kernels         (HIP: csrc/hip)
    shard          wavenumber-axis partition for multi-GPU jobs
    synth          synthetic inputs in the reference's file formats
"""
__all__ = ["host", "engine", "shard", "synth", "build"]
