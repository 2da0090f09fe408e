# Engine package (imported as ``xmc_gan_amd`` through the alias module at the repo root).
