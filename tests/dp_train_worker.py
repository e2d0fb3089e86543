"""Worker of tests/test_drivers_gpu.py::test_data_parallel_training_through_the_driver: one rank of
`torch.distributed.run ... dp_train_worker.py <config>` calling training.train()."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import avsi_amd  # noqa: E402,F401
from avsi_amd import training  # noqa: E402

if __name__ == '__main__':
    model = training.train(sys.argv[1])
    import torch.distributed as dist
    print('RANK %d STEPS %d' % (dist.get_rank(), model.global_step), flush=True)
    dist.barrier()
    dist.destroy_process_group()
