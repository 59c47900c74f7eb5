import sys, time
sys.path.insert(0, '/root/repo/lstm-unet_amd')
import torch
import Networks, Params
net = Params.CTCParams.net_kernel_params
m = Networks.ULSTMnet2D(net, 'NCHW', True, seed=0, precision=sys.argv[1] if len(sys.argv) > 1 else 'fp32')
frames = [torch.randn(1, 1, 1, 256, 256, device='cuda') for _ in range(4)]
for i in range(24): m(frames[i % 4], training=False)
torch.cuda.synchronize()
