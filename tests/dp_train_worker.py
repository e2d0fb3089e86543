"""Worker of tests/test_drivers_gpu.py::test_data_parallel_training_through_the_driver: one rank of
`torch.distributed.run ... dp_train_worker.py <config>` calling training.train()."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import avsi_amd  # noqa: E402,F401
from avsi_amd import training  # noqa: E402

if __name__ == '__main__':
    poison = os.environ.get('AVSI_TEST_NAN_RANK')
    if poison is not None and int(poison) == int(os.environ.get('RANK', '0')):
        # this rank's batches are poisoned (NaN masks -> NaN loss): every rank must leave at the same step
        plain = training.unpack_batch

        def poisoned(*a, **kw):
            feed, paths = plain(*a, **kw)
            feed['masks'] = feed['masks'] * float('nan')
            return feed, paths
        training.unpack_batch = poisoned
    park = os.environ.get('AVSI_TEST_PARK_RANK')
    if park is not None and int(park) == int(os.environ.get('RANK', '0')):
        # from this rank's second batch on, 16 of the GPU's 256 CUs are parked for 10 s with all their LDS taken (the stand-in
        # for another resident of the chip): the 32-way cooperative groups cannot have their XCD to themselves and time out,
        # every rank must fall back at the same step and go on
        import torch
        from avsi_amd import ops
        plain_unpack = training.unpack_batch
        state = {'n': 0}

        def parking(*a, **kw):
            state['n'] += 1
            if state['n'] == 2:
                state['release'] = torch.zeros(1, dtype=torch.int32, device='cuda')
                state['side'] = torch.cuda.Stream(priority=0)       # the trainer launches on a high-priority stream
                with torch.cuda.stream(state['side']):
                    ops.occupy_cus(16, state['release'], max_ms=10000)
            return plain_unpack(*a, **kw)
        training.unpack_batch = parking
    level2 = os.environ.get('AVSI_TEST_LEVEL2_RANK')
    if level2 is not None and int(level2) == int(os.environ.get('RANK', '0')):
        # this rank is on the batch-stationary kernels already (as after two fall-backs of its own inside validate()): a
        # timeout of a PEER must make it rewind and repeat with the others, not raise
        from avsi_amd import ops
        ops._COOP_FALLBACKS.extend(['test: preset', 'test: preset'])
    try:
        model = training.train(sys.argv[1])
    except SystemExit as e:
        open(os.path.join(os.path.dirname(sys.argv[1]), 'exit_rank%s' % os.environ.get('RANK', '0')), 'w').write(str(e.code))
        raise
    import torch.distributed as dist
    import hashlib
    print('RANK %d STEPS %d FALLBACKS %d VARS %s' % (dist.get_rank(), model.global_step, model.coop_fallbacks,
                                                    hashlib.sha1(model.variables.flat.cpu().numpy().tobytes()).hexdigest()),
          flush=True)
    dist.barrier()
    dist.destroy_process_group()
