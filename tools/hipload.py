import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
order = sys.argv[1]
if order == "torch_first":
    import torch
    print("torch sees", torch.cuda.is_available(), torch.cuda.device_count())
    from cafexp_amd import capi
    print("probe", capi.probe_fp64_mfma())
    x = torch.zeros(2, device="cuda:0"); print(x.cpu())
else:
    from cafexp_amd import capi
    print("probe", capi.probe_fp64_mfma())
    import torch
    print("torch sees", torch.cuda.is_available(), torch.cuda.device_count())
    x = torch.zeros(2, device="cuda:0"); print(x.cpu())
os.system("grep -E 'amdhip|hsa-runtime' /proc/%d/maps | awk '{print $6}' | sort -u" % os.getpid())
