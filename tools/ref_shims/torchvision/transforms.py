class _Stub:
    def __init__(self, *a, **k):
        pass

    def __call__(self, x):
        raise RuntimeError("torchvision stub: data pipeline is out of scope")


ToTensor = ToPILImage = Resize = RandomHorizontalFlip = CenterCrop = Lambda = Compose = _Stub
