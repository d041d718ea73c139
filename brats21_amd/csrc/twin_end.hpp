// closes namespace brats_f16 of the -DBRATS_FP16 twin build (twin_begin.hpp); no include guard: once per translation unit, last line
#ifdef BRATS_FP16
}  // namespace brats_f16
#endif
