import torch, numpy as np, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pacing_oracle as O
from tests import _golden as G
print('threads', torch.get_num_threads(), torch.__config__.parallel_info().split('\n')[0])
name='stride16'; target='backbone.dec_block3.conv_block.conv_layer1'
d=G.load(name); args=G.case_args(name)
for nt in (None, 1, 4):
    if nt: torch.set_num_threads(nt)
    O.TAP={}
    sd=G.to_state(G.sub(d,'init/'))
    _,og,_=O.train_step(sd,G.batch_of(d,0),0,args,True)
    k='backbone.dec_block3.conv_block.conv_layer1.conv.weight'
    print('threads', nt, 'oracle grad vs fixture:', G.rel_err(og[k].numpy(), d['step0/grad/'+k]))
    for gi,(z,y) in enumerate(O.TAP[target]):
        zz=z.detach().double(); dy=y.grad.double()
        mean=zz.mean((0,2,3),keepdim=True); var=zz.var((0,2,3),unbiased=False,keepdim=True); invstd=1/torch.sqrt(var+1e-5)
        xh=(zz-mean)*invstd
        g=torch.where(xh>0,dy,dy*0.01)
        n=zz.numel()/zz.shape[1]
        dz=invstd*(g-g.sum((0,2,3),keepdim=True)/n-xh*(g*xh).sum((0,2,3),keepdim=True)/n)
        print('  group',gi,'formula vs autograd z.grad: max abs',float((dz-z.grad.double()).abs().max()),'max|dz|',float(z.grad.abs().max()))
