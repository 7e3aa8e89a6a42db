import sys, os, ctypes
sys.path.insert(0, 'svt-av1-1_amd/python')
order = sys.argv[1]
if order == 'torch_first':
    import torch
    print('torch cuda avail', torch.cuda.is_available())
    x = torch.zeros(4, device='cuda:0')
    import svtav1_hip
    c = svtav1_hip.Context(0)
    print('ctx ok after torch')
else:
    import svtav1_hip
    c = svtav1_hip.Context(0)
    print('ctx ok')
    import torch
    print('torch cuda avail', torch.cuda.is_available())
    x = torch.zeros(4, device='cuda:0')
    print('torch ok after ctx')
os.system("grep -E 'amdhip64|hsa-runtime' /proc/%d/maps | awk '{print $6}' | sort -u" % os.getpid())
