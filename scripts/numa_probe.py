import os, glob, torch
p = torch.cuda.get_device_properties(0)
print([a for a in dir(p) if 'pci' in a.lower()])
for a in ('pci_domain_id','pci_bus_id','pci_device_id'):
    print(a, getattr(p,a,None))
for d in glob.glob('/sys/class/drm/card*/device'):
    try:
        print(d, os.path.realpath(d), open(d+'/numa_node').read().strip(), open(d+'/vendor').read().strip())
    except Exception as ex: print(d, ex)
print(os.sched_getaffinity(0).__len__())
