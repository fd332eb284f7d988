; bsgs_import.pbi -- PureBasic bindings of libbsgs_hip.so (include/bsgs_hip.h) for the host 1_9_7File.pb
; ("bsgscudaHT 1.9.7-file0").  PureBasic 5.31 x64, ASCII mode, thread-safe, as the reference is compiled (1_9_7File.pb:3-10).
;
; ROUTE A (no other source change): this file replaces the block  Import "lib\cuda.lib" ... EndImport  (1_9_7File.pb:55-106).
;   Every entry keeps the reference's name, parameter order and width (.i = pointer-sized integer, .s = ASCII string); the
;   return value is a CUresult-style integer, 0 = success (1_9_7File.pb:2195-2197).  All 48 names of the original block are
;   exported by the library, so the block links unchanged; v1.9.7 calls 23 of them (SURVEY.md 8b), the rest forward or answer
;   CUDA_ERROR_NOT_SUPPORTED.  On Windows build an import library from the DLL's export list and keep  Import "bsgs_hip.lib".
; ROUTE B: the native API below the marker; apply integration/route_b_1_9_7File.diff.
CompilerIf #PB_Compiler_OS = #PB_OS_Windows
  Import "bsgs_hip.lib"
CompilerElse
  ImportC "libbsgs_hip.so"
CompilerEndIf
  cuInit(Flags.i)
  cuMemGetInfo_v2(freebytes.i, totalbytes.i)            ; reports 40 % of the free HBM: the engine keeps its own layouts beside the host's buffer
  cuEventCreate(phEvent.i, Flags.i)
  cuEventDestroy(hEvent.i)
  cuEventQuery(hEvent.i)
  cuEventRecord(hEvent.i, Stream.i)
  cuEventSynchronize(hEvent.i)
  cuDeviceTotalMem(bytes.i, dev.i)
  cuDeviceTotalMem_v2(bytes.i, dev.i)
  cuDeviceComputeCapability(major.i, minor.i, dev.i)     ; gfx950 answers 9.5
  cuDeviceGetCount(count.i)
  cuDeviceGetName(name.s, len.i, dev.i)
  cuDeviceGetAttribute(pi.i, attrib.i, dev.i)            ; attrib 16 = compute units (256)
  cuDeviceGet(device.i, ordinal.i)
  cuGetErrorName(err.i, err_string.s)
  cuCtxCreate(pctx.i, flags.i, dev.i)
  cuCtxCreate_v2(pctx.i, flags.i, dev.i)
  cuMemAlloc(dptr.i, bytesize.i)
  cuMemAlloc_v2(dptr.i, bytesize.i)
  cuModuleGetGlobal(dptr.i, bytesize.i, hmodule.i, name.i)
  cuModuleGetGlobal_v2(dptr.i, bytesize.i, hmodule.i, name.i)   ; "_A": a 120-byte device-visible parameter block
  cuModuleLoadData(hmodule.i, image.i)                   ; the decoded PTX text is accepted and ignored: the HIP kernel is built in
  cuModuleLoad(hmodule.i, fname.i)
  cuModuleGetFunction(hfunc.i, hmod.i, name.s)           ; "_test1"
  cuParamSetSize(hfunc.i, numbytes.i)
  cuParamSetv(hfunc.i, offset.i, ptr.i, numbytes.i)
  cuParamSeti(hfunc.i, offset.i, value.i)
  cuFuncSetBlockShape(hfunc.i, x.i, y.i, z.i)
  cuLaunchGridAsync(hfunc.i, x.i, y.i, z.i, hstream.i)
  cuLaunchGrid(f.i, grid_width.i, grid_height.i)         ; one tile = 2*t*b*p giant steps
  cuFuncSetSharedSize(f.i, numbytes.i)
  cuFuncSetCacheConfig(f.i, config.i)
  cuLaunch(f.i)
  cuFuncGetAttribute(pi.i, attrib.i, f.i)
  cuStreamCreate(hStream.i, Flags.i)
  cuStreamCreate_v2(hStream.i, Flags.i)
  cuStreamDestroy(hStream.i)
  cuStreamSynchronize(hStream)
  cuStreamQuery(hStream.i)
  cuCtxSynchronize()
  cuMemcpyDtoH(dstHost.i, srcDevice.i, ByteCount.i)
  cuMemcpyDtoH_v2(dstHost.i, srcDevice.i, ByteCount.i)
  cuMemcpyHtoD(dstDevice.i, srcHost.i, ByteCount.i)
  cuMemcpyHtoD_v2(dstDevice.i, srcHost.i, ByteCount.i)
  cuMemFree(dptr.i)
  cuMemFree_v2(dptr.i)
  cuCtxDestroy(ctx.i)
  cuCtxDestroy_v2(ctx.i)

  ; ---- ROUTE B: native API (include/bsgs_hip.h).  int return: 0 = ok, negative = error, text from bsgs_last_error() ----
  bsgs_last_error()                                                  ; -> *ascii
  bsgs_dev_count(*n)
  bsgs_dev_open(id.l, *dev)
  bsgs_dev_close(dev.i)
  bsgs_dev_name(dev.i, *buf, len.l)
  bsgs_dev_meminfo(dev.i, *freebytes, *totalbytes)
  bsgs_upload_g2(dev.i, *image, t.l, b.l, p.l)                       ; *GiantArrPacked = the <t>_<b>_<p>_<w>_g2.BIN image verbatim
  bsgs_upload_htgpu(dev.i, *image, ht_items.q, w.q, layout.l)        ; *GpuHT = the ..._htGPUv0.BIN image verbatim; layout 0 = automatic
  bsgs_broadcast_tables(*devs, n.l)                                  ; devs(0) loaded as above -> replicas on the other GPUs over xGMI
  bsgs_step(dev.i, *px, *py, *hits, max_hits.l, *nhits)              ; one tile; hits = {code.l, idx.l} pairs like the legacy header
  bsgs_set_walk(dev.i, *p0_xy, *stride_xy)                           ; P0 = first GetJob centre, stride = PUBADDBIG: 64 bytes x||y each
  bsgs_run_walk(dev.i, first_tile.q, ntiles.l, *hits, max_hits.l, *nhits, *kernel_ms)   ; hits = {code.l, idx.l, tile.l, 0}
  bsgs_enqueue(dev.i, *centres, ntiles.l)
  bsgs_collect(dev.i, *hits, max_hits.l, *nhits, *kernel_ms)
  bsgs_set_flags(dev.i, flags.l)                                     ; 1 = reproduce the reference kernel's NEGMODP bug bit for bit
  bsgs_build_baby_tables(dev.i, w.q, htsz.l, *htgpu_out, *htcpu_out, install_layout.l)   ; GPU table builder: the two HT file images
  bsgs_generate_g2(dev.i, *addpubg_xy, t.l, b.l, p.l)
  bsgs_download_g2(dev.i, *image_out, bytes.i)
  bsgs_tiles_per_launch(dev.i, *n)                                   ; how many tiles one call of bsgs_run_walk should carry (the batch GetJob dispenses)
  bsgs_chain_placement(dev.i, *info5, *grade2)                       ; diagnostics: where the engine put its scratch (nothing to do for the host)
  bsgs_tune_placement(dev.i, candidates.l, *ms_out, *chosen2, *final_ms)   ; optional: choose the buffer placement by timed launches as well
  bsgs_prepare(dev.i)                                                ; round 3: allocate the scratch at start-up (like cuMemAlloc_v2 before the search loop, :2251)
  bsgs_build_baby_table_ext(dev.i, w.q, htsz.l, layout.l)            ; -w above 32: table built in GPU memory (layout 4 = 64-byte lines + overflow set); INTEGRATION.md
  bsgs_ext_overflow_capacity(w.q, htsz.l, layout.l, *ovf_cap)        ; multi-process hosts: size of the overflow set of such a table ...
  bsgs_alloc_table_ext_recv(dev.i, w.q, htsz.l, layout.l, *lines, *ovf, *ovf_cap)        ; ... receive buffers from the engine's allocator (reserved memory group above 40 GiB)
  bsgs_build_baby_table_ext_device(dev.i, w.q, htsz.l, layout.l, lines.i, ovf.i, ovf_cap.q, *ovf_n, *overflow_buckets)   ; ... rank 0 builds into its pair
  bsgs_install_table_ext_device(dev.i, lines.i, ovf.i, ovf_n.q, overflow_buckets.q, w.q, htsz.l, layout.l)                ; ... every rank installs its pair after the broadcast
  bsgs_compat_stats_ex(*launches, *served, *batches, *wasted_tiles)  ; route A: how the adaptive predicted batches fared
  bsgs_table_checksum(dev.i, *sums4)                                 ; round 4: 64-bit sums of lines / overflow set / image / giants, computed on the GPU: equal across replicas (compare after bsgs_broadcast_tables)
  bsgs_build_info()                                                  ; -> *ascii: the -D switches of the library; must be empty for a library that searches ("WRONG-RESULTS:..." = a timing experiment)
  bsgs_quirk_count(dev.i, *listed)                                   ; how many giants reference-quirk mode (bsgs_set_flags 1) re-computes after every launch
  bsgs_table_census(dev.i, *out8)                                    ; round 5: the table verified like checkHT / checkHTpack (:3599-3627, :3101-3134): out(7) = entries found must equal out(6) = w, out(4) = 0
  bsgs_table_lookup(dev.i, *keys64, n.q, *found)                     ; ... and sampled membership through the shipped probe: n 64-bit keys (low 64 bits of x(k*G)) in, n bytes out
  bsgs_sample_g2(dev.i, *idx64, n.l, *out_xy)                        ; round 6: n sampled giants as 64 bytes x_le || y_le each -- compare with (idx + 1) * ADDPUBG like checkGiantArr (:1524-1559, called :1941)
  bsgs_broadcast_tables_ex(*devs, n.l, transport.l, what.l, *transport_used, *seconds)   ; replicas with the transport chosen (0 auto, 1 RCCL over xGMI, 2 peer copies) and reported; what: 1 giants | 2 table
  bsgs_startup_ext_tables(*devs, n.l, w.q, htsz.l, layout.l, strategy.l, transport.l, *report)   ; -w above 32 on several GPUs: 0 = GPU 0 builds + broadcast, 1 = every GPU builds its own (default), 2 = 1/N each + all-gather
  bsgs_version()                                                     ; -> *ascii: the library's version line (the host's banner)
  bsgs_table_info(dev.i, *layout, *device_bytes, *overflow_buckets)  ; what is installed: layout code, bytes on the device, over-full buckets (the "Extended table: ..." line)
  bsgs_set_tiles_per_launch(dev.i, n.l)                              ; short jobs (-infile over a small range): launches of n tiles, scratch sized for them (0 = the engine's default)
  bsgs_engine_geometry(dev.i, *threads, *giants_per_thread)          ; the engine's own thread x batch factorisation of t*b*p (only the hit index is visible outside)
  bsgs_share_tables(owner.i, twin.i)                                 ; two engines on ONE GPU (two public keys searched side by side): the twin probes the owner's table in place; free the twin first
EndImport
