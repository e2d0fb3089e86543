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
    try:
        model = training.train(sys.argv[1])
    except SystemExit as e:
        open(os.path.join(os.path.dirname(sys.argv[1]), 'exit_rank%s' % os.environ.get('RANK', '0')), 'w').write(str(e.code))
        raise
    import torch.distributed as dist
    print('RANK %d STEPS %d' % (dist.get_rank(), model.global_step), flush=True)
    dist.barrier()
    dist.destroy_process_group()
