from .network import (ResNetActorBase, ResNetActor_ADMM, ResNetActor_HQS, ResNetActor_PG, ResNetActor_APG,  # noqa: F401
                      ResNetActor_RED, ResNetActor_IADMM, ResNetActor_AMP, ResNetActor_SPI)
