import torch, sys, os
sys.path.insert(0, os.getcwd())
from pacingpseudo_amd._lib import lib
dev='cuda'; st=torch.cuda.current_stream().cuda_stream
B,H,W,C,N=2,32,32,32,32
x=torch.zeros(B,H,W,C,device=dev)
ys,xs=torch.meshgrid(torch.arange(H),torch.arange(W),indexing='ij')
for b in range(B):
    x[b,:,:,0]=(ys*100+xs+1000*b).float().to(dev)      # channel 0 = pixel id
    x[b,:,:,1]=1.0
w=torch.zeros(N,C,3,3); w[0,0,1,1]=1.0; w[1,0,1,2]=1.0; w[2,0,2,1]=1.0; w[3,1,1,1]=1.0
wf=torch.zeros(N,9,C,device=dev)
lib.pp_pack_conv3x3_weights_f16x3(w.to(dev).data_ptr(),N,C,C,wf.data_ptr(),None,st)
out=torch.full((B,H,W,N),7.0,device=dev)
lib.pp_conv3x3_fwd_f16x3(x.data_ptr(),C,C,wf.data_ptr(),None,out.data_ptr(),N,N,B,H,W,1,0,None,st)
torch.cuda.synchronize()
o=out.cpu()
print('ch0 (identity) img0 rows 0..5, cols 0..11'); print(o[0,:6,:12,0])
print('ch1 (right neighbour)'); print(o[0,:3,:12,1])
print('ch2 (below)'); print(o[0,:6,:6,2])
print('ch3 (ones)'); print(o[0,:3,:12,3])
print('img1 ch0'); print(o[1,:6,:12,0])
print('ch5 (should be 0)', o[...,5].abs().max())
