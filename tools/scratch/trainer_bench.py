"""Per-iteration wall time of nnUNetTrainer_simple.run_iteration at the benchmarked shape (the product path: host batch ->
pinned upload -> engine step -> loss scalar back), against the bare engine loop bench.py times (diagnostic)."""
import os, sys, time, random, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from e2enet_medical_amd.training.network_training.nnUNetTrainer_simple import nnUNetTrainer_simple
from e2enet_medical_amd.training.network_training.sparselearning.core_channel import Masking, CosineDecay
plans = {'plans_per_stage': {0: {'batch_size': 2, 'patch_size': [128, 128, 128], 'num_pool_per_axis': [5, 5, 5],
                                 'pool_op_kernel_sizes': [[2, 2, 2]] * 5, 'conv_kernel_sizes': [[3, 3, 3]] * 6, 'do_dummy_2D_data_aug': False}},
         'base_num_features': 32, 'num_modalities': 4, 'num_classes': 3, 'all_classes': [1, 2, 3],
         'transpose_forward': [0, 1, 2], 'transpose_backward': [0, 1, 2], 'conv_per_stage': 2}
tr = nnUNetTrainer_simple(plans, 0, output_folder="/tmp/tb", batch_dice=True, Tconv='shiftConvPP', max_num_epochs=1, num_batches_per_epoch=1)
tr.base_num_features_override = 32
torch.manual_seed(0)
tr.synthetic_data = True
net, opt = tr.initialize(True)
class A: adv = False; fix = False; update_frequency = 1200; final_density = 0.05
random.seed(0)
mask = Masking(opt, death_rate=0.5, death_mode='magnitude', death_rate_decay=CosineDecay(0.5, 1000), growth_mode='random', redistribution_mode='none', args=A())
mask.add_module(net, sparse_init='uniform', density=0.2)
batch = next(tr.tr_gen)
def gen_host():
    while True: yield {'data': batch['data'], 'target': batch['target']}
dev_batch = {'data': batch['data'].cuda(), 'target': [t.cuda() for t in batch['target']]}
def gen_dev():
    while True: yield dev_batch
pinned = {'data': batch['data'].pin_memory(), 'target': [t.pin_memory() for t in batch['target']]}
def gen_pin():
    while True: yield pinned
for name, g in (("device-resident batch", gen_dev()), ("pinned host batch", gen_pin()), ("pageable host batch", gen_host())):
    for _ in range(3): tr.run_iteration(g, True, False, mask)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): tr.run_iteration(g, True, False, mask)
    torch.cuda.synchronize()
    print("%-24s %.2f ms / iteration" % (name, (time.perf_counter() - t0) / 10 * 1e3))
