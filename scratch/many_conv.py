import sys; sys.path.insert(0,"/root/repo")
import torch
from semantic_pyramid_for_image_generation_amd import ops
cin,cout,hw,B=[int(a) for a in sys.argv[1:5]]
dt=torch.bfloat16
x=ops.nhwc_empty(B,cin,hw,hw,dt,'cuda'); x.normal_()
w=(torch.randn(cout*9*cin,device='cuda')*0.05).to(dt)
bias=torch.randn(cout,device='cuda')
y=ops.nhwc_empty(B,cout,hw,hw,dt,'cuda')
for _ in range(60):
    ops.conv_launch(x,w.data_ptr(),bias,y,None,None,None,0.0,B,hw,hw,cin,cout,cout,3,1,dt)
torch.cuda.synchronize()
