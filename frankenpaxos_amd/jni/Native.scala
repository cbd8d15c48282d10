// Native.scala -- the reference-side binding a frankenpaxos maintainer would add (source only: no
// JDK / scalac in this image).  Drop into jvm/src/main/scala/frankenpaxos/gpu/ of the reference.
//
// `Native` is the JNI surface of frankenpaxos_amd/jni/fpx_jni.c; `GpuPhase2` is a batched stand-in for
// the acceptors of every group plus one proxy leader that plugs into the unchanged Actor/Transport
// trait surface: it is an Actor whose `receive` only ENQUEUES the decoded Phase2a / Phase2b messages,
// and a zero-delay Transport timer (the "tick") flushes the queue through ONE native call and then
// `send`s the Phase2b / Nack / Chosen messages the Scala handlers would have sent
// (multipaxos/Acceptor.scala:192-219, multipaxos/ProxyLeader.scala:246-253).
package frankenpaxos.gpu

import frankenpaxos.Actor
import frankenpaxos.Chan
import frankenpaxos.Logger
import frankenpaxos.multipaxos._
import frankenpaxos.roundsystem.RoundSystem
import scala.collection.mutable

object Native {
  System.loadLibrary("fpxjni") // libfpxjni.so -> libfpx.so

  // status codes of include/fpx.h
  val OK = 0; val EINVAL = 1; val EFATAL_UNKNOWN_SLOTROUND = 2

  @native def create(cfg: Array[Int]): Long // < 0: -status
  @native def destroy(handle: Long): Int
  @native def acceptorPhase2a(handle: Long, n: Int, slot: Array[Int], round: Array[Int],
                              value: Array[Int], targetMask: Array[Long], voteBits: Array[Long],
                              nackBits: Array[Long], nackRound: Array[Int]): Int
  @native def proxyOpen(handle: Long, n: Int, slot: Array[Int], round: Array[Int],
                        value: Array[Int], isNew: Array[Byte]): Int
  @native def proxyPhase2b(handle: Long, n: Int, slot: Array[Int], round: Array[Int],
                           voteBits: Array[Long], newlyChosen: Array[Byte],
                           chosenRound: Array[Int], chosenValue: Array[Int]): Int
  @native def phase2Fused(handle: Long, n: Int, slot: Array[Int], round: Array[Int],
                          value: Array[Int], targetMask: Array[Long], chosen: Array[Byte],
                          chosenRound: Array[Int], chosenValue: Array[Int], nackRound: Array[Int]): Int
  @native def quorumEval(cfg: Array[Int], n: Int, nodes: Array[Long], strict: Int,
                         out: Array[Byte]): Int

  // page-locked batches: direct buffers over fpx_host_alloc memory (fill through asIntBuffer /
  // asLongBuffer views in ByteOrder.nativeOrder), DMA'd without pinning or copying
  @native def hostAlloc(bytes: Long): java.nio.ByteBuffer // null on failure
  @native def hostFree(buffer: java.nio.ByteBuffer): Int
  @native def phase2FusedDirect(handle: Long, n: Int, slot: java.nio.ByteBuffer,
                                round: java.nio.ByteBuffer, value: java.nio.ByteBuffer,
                                targetMask: java.nio.ByteBuffer, chosen: java.nio.ByteBuffer,
                                chosenRound: java.nio.ByteBuffer, chosenValue: java.nio.ByteBuffer,
                                nackRound: java.nio.ByteBuffer): Int

  // the rows around the fused step
  @native def roundLeader(numLeaders: Int, round: Int): Int // < 0: -status (numLeaders < 1)
  @native def acceptorPhase1a(handle: Long, group: Int, round: Int, chosenWatermark: Int,
                              targetMask: Array[Long], bits: Array[Long]): Int
  @native def leaderPhase1bScan(handle: Long, chosenWatermark: Int, numGroups: Int,
                                quorumMasks: Array[Long], cap: Int, maxSlot: Array[Int],
                                safeRound: Array[Int], safeValue: Array[Int]): Int
  @native def replicaChosen(handle: Long, n: Int, slot: Array[Int], value: Array[Int],
                            mask: Array[Byte], state: Array[Int]): Int
  @native def replicaChosenNoopRange(handle: Long, slotStart: Int, slotEnd: Int,
                                     state: Array[Int]): Int
  // Mencius noop ranges: n per call (the fused step), and the unfused pieces for one range
  @native def noopRangesFused(handle: Long, n: Int, numGroups: Int, slotStart: Array[Int],
                              slotEnd: Array[Int], round: Array[Int], targetMasks: Array[Long],
                              voteBits: Array[Long], nackBits: Array[Long], nackRound: Array[Int],
                              isNew: Array[Byte], chosen: Array[Byte]): Int
  @native def acceptorPhase2aNoopRange(handle: Long, slotStart: Int, slotEnd: Int, round: Int,
                                       numGroups: Int, targetMasks: Array[Long],
                                       bits: Array[Long], nackRound: Array[Int]): Int
  @native def proxyOpenNoopRange(handle: Long, slotStart: Int, slotEnd: Int, round: Int,
                                 isNew: Array[Byte]): Int
  @native def proxyPhase2bNoopRange(handle: Long, slotStart: Int, slotEnd: Int, round: Int,
                                    numGroups: Int, voteBits: Array[Long],
                                    newlyChosen: Array[Byte]): Int
  @native def epxCreate(numReplicas: Int, numKeys: Int, device: Int): Long // < 0: -status
  @native def epxDestroy(handle: Long): Int
  @native def epxPreaccept(handle: Long, m: Int, numReplicas: Int, leader: Array[Int],
                           number: Array[Int], key: Array[Int], isSet: Array[Byte],
                           respMask: Array[Byte], seenMask: Array[Byte], rank: Array[Int],
                           fast: Array[Byte], deps: Array[Int], leaderDeps: Array[Int],
                           ownValuesEnd: Array[Int]): Int
  // EPaxos on the command log (Replica.cmdLog): Prepare, Accept, handlePreAccept with ballots / Nacks / re-sent
  // replies.  replies packs the per-message reply bit sets back to back (m bytes each), see fpx_jni.c
  @native def epxCreateWithLog(numReplicas: Int, numKeys: Int, device: Int, numInstances: Int): Long
  @native def epxPrepare(handle: Long, m: Int, numReplicas: Int, leader: Array[Int], number: Array[Int],
                         ballotOrdering: Array[Int], ballotReplica: Array[Int], targetMask: Array[Byte],
                         replies: Array[Byte], nackBallot: Array[Int], prepareOk: Array[Int]): Int
  @native def epxAccept(handle: Long, m: Int, leader: Array[Int], number: Array[Int],
                        ballotOrdering: Array[Int], ballotReplica: Array[Int], tripleId: Array[Int],
                        key: Array[Int], isSet: Array[Byte],
                        targetMask: Array[Byte], replies: Array[Byte], nackBallot: Array[Int]): Int
  @native def epxHandlePreaccept(handle: Long, m: Int, numReplicas: Int, leader: Array[Int],
                                 number: Array[Int], ballotOrdering: Array[Int],
                                 ballotReplica: Array[Int], key: Array[Int], isSet: Array[Byte],
                                 tripleId: Array[Int], depsIn: Array[Int], depsInValuesEnd: Array[Int],
                                 targetMask: Array[Byte], replies: Array[Byte], nackBallot: Array[Int],
                                 replyDeps: Array[Int], replyEndTriple: Array[Int]): Int
  @native def epxReadCmdlog(handle: Long, numReplicas: Int, replica: Int, leader: Int, number: Int,
                            entry: Array[Int]): Int
  // multi-GPU: one context per GPU, one RCCL communicator over them (fpx_comm_*); the 128-byte id of
  // commUniqueId travels to the other ranks over the actors' own transport
  @native def commUniqueId(id: Array[Byte]): Int
  @native def commCreate(handle: Long, id: Array[Byte], rank: Int, world: Int): Int
  @native def commDestroy(handle: Long): Int
  // wire adapter: a tick of ProxyLeaderInbound byte arrays packed into one direct buffer + n + 1
  // offsets -> fields = kind | slot | round | isNoop | valueLen | groupIndex | acceptorIndex (7 x n)
  @native def wireDecodeProxyLeaderInbound(buf: java.nio.ByteBuffer, offsets: Array[Long], n: Int,
                                           fields: Array[Int], valueOff: Array[Long],
                                           badIndex: Array[Int]): Int

  def check(status: Int, logger: Logger): Unit = status match {
    case OK                       => ()
    case EINVAL                   => throw new IllegalArgumentException("libfpx: require failed")
    case EFATAL_UNKNOWN_SLOTROUND =>
      logger.fatal("A ProxyLeader received a Phase2b in a slot and round it never sent a Phase2a in.")
    case s => logger.fatal(s"libfpx status $s")
  }
}

// A GPU-backed replacement for the acceptor groups + one proxy leader of a (non-flexible or
// flexible) MultiPaxos deployment.  Leaders keep sending Phase2a to it exactly as they send to a
// ProxyLeader (multipaxos/Leader.scala:364-398); replicas keep receiving Chosen from it.
class GpuPhase2[Transport <: frankenpaxos.Transport[Transport]](
    address: Transport#Address,
    transport: Transport,
    logger: Logger,
    config: Config[Transport],
    numSlots: Int = 1 << 20
) extends Actor(address, transport, logger) {
  config.checkValid()
  override type InboundMessage = ProxyLeaderInbound
  override val serializer = ProxyLeaderInboundSerializer

  private val perGroup = config.acceptorAddresses(0).size
  private val cfg: Array[Int] =
    if (!config.flexible)
      Array(numSlots, perGroup, config.numAcceptorGroups, 1, config.f, /*THRESHOLD*/ 0, 0, 0,
            config.numLeaders, /*ACCEPTOR*/ 0, 4, 0, 0, 0, 0)
    else
      Array(numSlots, config.numAcceptorGroups * perGroup, 1, 1, config.f, /*GRID*/ 2,
            config.numAcceptorGroups, perGroup, config.numLeaders, 0, 4, 0, 0, 0, 0)
  private val handle = Native.create(cfg)
  if (handle < 0) Native.check((-handle).toInt, logger)

  private val roundSystem = new RoundSystem.ClassicRoundRobin(config.numLeaders)
  private val leaders = for (a <- config.leaderAddresses) yield chan[Leader[Transport]](a, Leader.serializer)
  private val replicas = for (a <- config.replicaAddresses) yield chan[Replica[Transport]](a, Replica.serializer)

  // value ids: the int32 the GPU carries stands for a CommandBatchOrNoop kept on the JVM side
  private val values = mutable.Buffer[CommandBatchOrNoop]()
  private val pending = mutable.Buffer[Phase2a]()

  // one tick: a zero-delay timer, i.e. "after the messages already queued on the event loop"
  private val tick = timer("gpuPhase2Tick", java.time.Duration.ZERO, () => flushTick())

  override def receive(src: Transport#Address, inbound: ProxyLeaderInbound): Unit = {
    inbound.request match {
      case ProxyLeaderInbound.Request.Phase2A(p) =>
        if (pending.isEmpty) tick.start()
        pending += p
      case ProxyLeaderInbound.Request.Phase2B(_) =>
        logger.fatal("GpuPhase2 tallies on the device; it never receives Phase2b messages.")
      case ProxyLeaderInbound.Request.Empty =>
        logger.fatal("Empty ProxyLeaderInbound encountered.")
    }
  }

  private def flushTick(): Unit = {
    val n = pending.size
    val slot = new Array[Int](n); val round = new Array[Int](n); val value = new Array[Int](n)
    for ((p, i) <- pending.zipWithIndex) {
      slot(i) = p.slot; round(i) = p.round
      value(i) = values.size; values += p.commandBatchOrNoop
    }
    val chosen = new Array[Byte](n); val cr = new Array[Int](n); val cv = new Array[Int](n)
    val nr = new Array[Int](n)
    // dense delivery (targetMask = null).  A thrifty deployment passes one random f+1 / grid-column
    // mask per message here (ProxyLeader.scala:190-196).
    Native.check(Native.phase2Fused(handle, n, slot, round, value, null, chosen, cr, cv, nr), logger)
    for (i <- 0 until n) {
      if (chosen(i) != 0) {
        // ProxyLeader.scala:246-253
        val msg = ReplicaInbound().withChosen(Chosen(slot = slot(i), commandBatchOrNoop = values(cv(i))))
        replicas.foreach(_.send(msg))
      }
      if (nr(i) >= 0) {
        // Acceptor.scala:197-198: Nack(round = acceptor's round) to leaders(roundSystem.leader(phase2a.round))
        leaders(roundSystem.leader(round(i))).send(LeaderInbound().withNack(Nack(round = nr(i))))
      }
    }
    pending.clear()
  }
}
