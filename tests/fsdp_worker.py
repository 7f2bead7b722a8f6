"""One rank of the 2-rank FSDP FULL_SHARD test (tests/test_gpu_features.py::test_two_rank_fsdp_full_shard_with_checkpointing).
Test infrastructure: started as a fresh child process per rank; both ranks use cuda:0 and a gloo process group (RCCL refuses two
ranks on one device; FSDP's all-gather / reduce-scatter run over gloo).

    RANK=r WORLD_SIZE=2 MASTER_PORT=p python tests/fsdp_worker.py <eager|ours|ours_wcache> <out.pt>

The reference's real run is `--fsdp "full_shard auto_wrap"` around each decoder layer with gradient checkpointing and bf16 autocast
(run_train.sh:17-18,:36,:42-43, utils/kd_trainer.py:244): every weight the quantizers see is a view into a flat parameter that is
all-gathered before the layer runs and freed after it, also for the checkpoint recompute."""
import functools
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    impl, out = sys.argv[1], sys.argv[2]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{os.environ['MASTER_PORT']}", rank=rank, world_size=world)
    from torch.distributed.algorithms._checkpoint.checkpoint_wrapper import CheckpointImpl, apply_activation_checkpointing, checkpoint_wrapper
    from torch.distributed.fsdp import FullyShardedDataParallel as FSDP, ShardingStrategy
    from torch.distributed.fsdp.wrap import transformer_auto_wrap_policy

    import llm_qat_amd
    import llm_qat_amd.utils_quant as UQ
    import tiny_llama as TL
    quant = TL.EagerQuant() if impl == "eager" else UQ
    llm_qat_amd.enable_weight_quant_cache(impl == "ours_wcache")
    model = TL.load_deterministic(TL.TinyLlama(quant, w_bits=4, a_bits=8, kv_bits=4).bfloat16()).cuda()
    apply_activation_checkpointing(model, checkpoint_wrapper_fn=functools.partial(checkpoint_wrapper, checkpoint_impl=CheckpointImpl.NO_REENTRANT),
                                   check_fn=lambda m: isinstance(m, TL.Layer))
    fsdp = FSDP(model, auto_wrap_policy=functools.partial(transformer_auto_wrap_policy, transformer_layer_cls={TL.Layer}),
                sharding_strategy=ShardingStrategy.FULL_SHARD, device_id=0, use_orig_params=True)
    opt = torch.optim.SGD(fsdp.parameters(), lr=0.05)
    ids = TL.deterministic_batch(seed=7 + rank).cuda()   # each rank its own batch, the same for every implementation
    losses = []
    for _ in range(3):
        opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss, _ = fsdp(ids, labels=ids)
        loss.backward()
        opt.step()
        losses.append(loss.detach().float().cpu())
    with FSDP.summon_full_params(fsdp):
        params = {n.replace("_fsdp_wrapped_module.", "").replace("_checkpoint_wrapped_module.", ""): p.detach().float().cpu().clone() for n, p in fsdp.named_parameters()}
    if rank == 0:
        torch.save({"losses": torch.stack(losses), "params": params}, out)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
