"""mm2chain: MI355X-native chaining DP (mm_chain_dp hot path of kisarur/minimap2-fpga) behind the reference's
dispatch surface.  The compute lives in libmm2chain_hip.so (HIP, gfx950); this package is the host-side mirror."""
from ._native import Params, Mm2cError, LIB_PATH, MM2C_F_IGNORE_SEG, MM2C_F_FORCE_GENERAL, load
from . import params, synth, sharding, stream
from .batch import (device_identity, last_host_variant, init, split_model, init_devices, device_count, split_tasks, shutdown, device_info, tune, ChainPlan, chain_batch_host, chain_batch_host_into, PinnedArray, chain_task, hardware_init, cleanup,
                    run_chaining_on_hw, mm_chain_dp, mm_chain_dp_batch, chain_epilogue_host, SeedPlan, seed_hits_batch, seed_chain_batch, MATCH_DTYPE,
                    HitPool, seed_chain_batch_pool, stage_stats, slot_stats, chain_task_pred)

__all__ = ["Params", "Mm2cError", "LIB_PATH", "MM2C_F_IGNORE_SEG", "MM2C_F_FORCE_GENERAL", "load", "params", "synth",
           "sharding", "stream", "device_identity", "last_host_variant", "init", "split_model", "init_devices", "device_count", "split_tasks", "shutdown", "device_info", "tune", "ChainPlan", "chain_batch_host", "chain_batch_host_into", "PinnedArray", "chain_task", "hardware_init",
           "cleanup", "run_chaining_on_hw", "mm_chain_dp", "mm_chain_dp_batch", "chain_epilogue_host", "SeedPlan", "seed_hits_batch", "seed_chain_batch", "MATCH_DTYPE",
           "HitPool", "seed_chain_batch_pool", "stage_stats", "slot_stats", "chain_task_pred"]
